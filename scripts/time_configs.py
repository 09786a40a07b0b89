#!/usr/bin/env python3
"""Times BASELINE.json configs[2] (vocoder) and configs[3] (1000 stereo streams, N = 8192, P = 128, autotune) at full size on one GPU.

These are parity-test shapes, not bench lines (bench.py measures configs[1]); this script records what they cost so that DESIGN.md can
state it.  Usage (GPU box):  python scripts/time_configs.py [c3] [c4] [--steps K] > gpurun_out/configs.json
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def carrier_torch(S, n, sr, device, kind="saw"):
    import torch
    t = torch.arange(n, dtype=torch.float64, device=device) / sr
    f = 110.0 * (1.0 + 0.01 * torch.arange(S, dtype=torch.float64, device=device)).unsqueeze(1)
    ph = torch.remainder(f * t, 1.0)
    x = (2.0 * ph - 1.0) if kind == "saw" else torch.sin(2 * math.pi * ph)
    return (0.5 * x).to(torch.float32).contiguous()


def run(name, cfg_kw, S, seconds, steps, with_carrier):
    import torch
    import bench
    from cpp_audio_amd import api
    dev = torch.device("cuda", 0)
    sr = 48000
    n = int(seconds * sr)
    voice = [bench.synth_inputs_torch(S, n, sr, 1234 + 7 * k, dev) for k in range(2)]
    car = carrier_torch(S, n, sr, dev) if with_carrier else None
    out = torch.empty(S, n, 2, dtype=torch.float32, device=dev)
    cfg = api.default_config(**cfg_kw)
    res = {"config": name, "streams": S, "seconds": seconds, "steps": steps}
    with api.Context(cfg, S, 0) as ctx:
        ctx.set_hip_stream(torch.cuda.current_stream().cuda_stream)
        cp = car.data_ptr() if car is not None else 0
        for k in range(3):  # warm-up: both scratch sets, steady-state frame counts
            ctx.process_device(voice[k % 2].data_ptr(), cp, n, out.data_ptr(), overlap=True)
        ctx.join()
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            ctx.process_device(voice[k % 2].data_ptr(), cp, n, out.data_ptr(), overlap=True)
        ctx.join()
        ctx.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res["ms_per_step"] = dt * 1e3
        res["frames_per_s"] = S * n / dt
        res["x_realtime_per_stream"] = n / sr / dt
        # per-kernel split from HIP events (a separate pass: the events serialise nothing but add launches)
        ctx.set_profiling(True)
        before, _ = ctx.kernel_ms()
        for k in range(2):
            ctx.process_device(voice[k % 2].data_ptr(), cp, n, out.data_ptr(), overlap=True)
        ctx.join()
        ctx.synchronize()
        after, _ = ctx.kernel_ms()
        res["kernel_ms_per_step"] = {k: round((after[k] - before[k]) / 2, 3) for k in after}
        st = ctx.stats()
        res["error_flags"] = int(st.error_flags)
        res["dropped_note_on"] = int(st.dropped_note_on)
        res["out_rms"] = float(out.double().pow(2).mean().sqrt())
    return res


def main():
    which = [a for a in sys.argv[1:] if a in ("c3", "c4")] or ["c3", "c4"]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
    c4_streams = int(sys.argv[sys.argv.index("--c4-streams") + 1]) if "--c4-streams" in sys.argv else 2000
    parity = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
    results = []
    if "c3" in which:
        kw = dict(parity, analysis_volume=0.0, voice_volume=0.0, carrier_volume=0.0, vocoder_volume=1.0, vocoder_count_bands=32,
                  vocoder_min_freq=100.0, vocoder_max_freq=20000.0, vocoder_modulator_window_size_seconds=0.10, vocoder_stride_seconds=0.005)
        results.append(run("configs[2]: vocoder only, 32 bands, S = 241, modulator FFT 8192, carrier FFT 512", kw, 256, 10.0, steps, True))
        kw2 = dict(kw, analysis_volume=1.0)
        results.append(run("configs[2] + the analysis / sine bank of configs[1] beside it", kw2, 256, 10.0, steps, True))
    if "c4" in which:
        kw = dict(parity, window_size=8192, stride=2048, n_channels=128, use_autotune=1, autotune_type=0, autotune_scale_mode=0, autotune_root_note=0)
        results.append(run("configs[3]: 1000 stereo = 2000 mono streams, N = 8192, H = 2048, P = 128, autotune major / C", kw, c4_streams, 10.0, steps, False))
    for r in results:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
