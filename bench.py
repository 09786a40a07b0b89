#!/usr/bin/env python3
"""Headline benchmark: audio frames/s of the rt.resynth analysis -> resynthesis hot path on MI355X.

One "step" = one pass of the whole hot path (analysis + tracking + sine-bank synthesis) over one batch of
synthetic input: BASELINE.json configs[1] = 256 mono streams x 10 s @ 48 kHz, 4096-pt window, 75 % overlap
(H = 1024), 64 partials.  Consecutive steps continue the same streams (steady state);
inputs are resident in HBM before the timed region.  N > 1: every rank runs its own 256 streams on its own
GPU (streams are independent: no collective on the data path), value = all ranks' frames / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per audio frame = 4 B float in + 2 x 4 B float out
ALGO_BYTES_PER_FRAME = 12.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # nominal vector fp64
# One fp64 VALU instruction of a wave64 occupies its SIMD for 4 cycles (16 lanes / clk: the 78.6 TFLOP/s above); at 2.4 GHz
# the 1024 SIMDs issue at most 614 G wave-instructions / s.  scripts/ubench/fp64_rate.hip measures 2.3 - 2.4 ns per instruction
# and SIMD with >= 2 waves resident (~70 % of nominal) and 4.4 ns with one wave.
VALU_PEAK_GINST = 1024 * 2.4 / 4.0


def algo_flops_per_frame(N, H, P):
    """FFT 5 N log2 N / H + window/|X|^2/peaks 12 N / H + sine bank 30 P."""
    import math
    return 5.0 * N * math.log2(N) / H + 12.0 * N / H + 30.0 * P


def host_cores():
    """Host threads we may actually use: affinity mask, clipped by the cgroup CPU quota when there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                cores = max(1, min(cores, quota // period))
        except Exception:
            pass
    cores = min(cores, 64)  # a bounded sample cannot feed more threads than this anyway
    env = os.environ.get("RA_BENCH_CPU_THREADS")
    return int(env) if env else cores


def measured_counters(kernel, S, n, N, H, P):
    """Per-step PMC sums of `kernel` from the committed rocprofv3 passes (profiles/r*/traffic.json: separate --pmc runs of FETCH_SIZE,
    WRITE_SIZE and SQ_INSTS_VALU) — PMC counters cannot be read from inside this process.  None unless a profile of exactly
    this workload shape exists."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
            w = doc["workload"]
            if (w["streams"], w["frames_per_stream"], w["window"], w["stride"], w["partials"]) != (S, n, N, H, P):
                continue
            return doc["kernels"][kernel], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def synth_inputs_torch(n_streams, n_frames, sr, seed, device):
    """Seeded synthetic streams of the SURVEY.md §8(d) recipe, generated on the GPU (float32, peak 0.5)."""
    import math
    import torch
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    K = 24

    def U(lo, hi, *shape):
        return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).to(device)

    x = torch.zeros(n_streams, n_frames, dtype=torch.float64, device=device)
    t = torch.arange(n_frames, dtype=torch.float64, device=device) / sr
    f0 = torch.exp(U(math.log(80.0), math.log(8000.0), n_streams, K))
    vib_rate = U(0.5, 6.0, n_streams, K)
    vib_depth = U(0.0, 0.3, n_streams, K)
    amp = 10.0 ** (U(-40.0, -6.0, n_streams, K) / 20.0)
    ph = U(0.0, 2 * math.pi, n_streams, K)
    vph = U(0.0, 2 * math.pi, n_streams, K)
    period = U(1.0, 3.0, n_streams, K)
    duty = U(0.5, 0.95, n_streams, K) * period
    gph = U(0.0, 3.0, n_streams, K)
    ramp = 0.01
    for k in range(K):
        f = f0[:, k:k + 1] * 2.0 ** (vib_depth[:, k:k + 1] / 12.0 * torch.sin(2 * math.pi * vib_rate[:, k:k + 1] * t + vph[:, k:k + 1]))
        phase = ph[:, k:k + 1] + 2 * math.pi * torch.cumsum(f, dim=1) / sr
        u = torch.remainder(t + gph[:, k:k + 1], period[:, k:k + 1])
        gate = torch.clamp(torch.minimum(u, duty[:, k:k + 1] - u) / ramp, 0.0, 1.0)
        x += amp[:, k:k + 1] * gate * torch.sin(phase)
    gn = torch.Generator(device=device)
    gn.manual_seed(seed + 1)
    x += 10.0 ** (-60.0 / 20.0) * torch.randn(n_streams, n_frames, generator=gn, device=device, dtype=torch.float64)
    x *= 0.5 / x.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    return x.to(torch.float32).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU (configs[1]: 256)")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds per stream per step")
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--stride", type=int, default=1024)
    ap.add_argument("--partials", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream-ordered", action="store_true", help="issue every block with ra_process_device (no overlap between consecutive blocks)")
    ap.add_argument("--cpu-streams", type=int, default=0, help="streams in the CPU baseline sample (0 = all)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from cpp_audio_amd import api

    sr = 48000
    S, n = args.streams, int(round(args.seconds * sr))
    N, H, P = args.window, args.stride, args.partials
    voice = synth_inputs_torch(S, n, sr, 1234 + 100003 * rank, dev)
    out = torch.empty(S, n, 2, dtype=torch.float32, device=dev)
    cfg = api.default_config(sample_rate=sr, window_size=N, stride=H, n_channels=P, analysis_volume=1.0, stereo_spread=0.0,
                             carrier_volume=0.0, seed=1234 + 100003 * rank)
    ctx = api.Context(cfg, S, local_rank)
    ctx.set_hip_stream(torch.cuda.current_stream().cuda_stream)
    # set-up, not a step: a 2 s block makes HIP load the code objects and creates the context's internal streams and events, then the
    # streams are put back to their initial state; the per-block scratch (GBs) is still allocated by the first warm-up step
    ctx.process_device(voice.data_ptr(), 0, min(n, 2 * sr), out.data_ptr())
    ctx.synchronize()
    ctx.reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Steps are consecutive blocks of the same streams.  They are issued like an offline job issues them (ra_process_device_async): the
    # caller's stream is not made to wait between blocks, so block k + 1's analysis and tracking overlap block k's rendering; every block
    # still does all of its work inside the timed region, which ends with a join + device synchronisation.  --stream-ordered issues each
    # block with ra_process_device instead (the stream waits for every block: nothing of block k + 1 starts before block k is done).
    def step():
        ctx.process_device(voice.data_ptr(), 0, n, out.data_ptr(), overlap=not args.stream_ordered)

    for _ in range(args.warmup):
        step()
    ctx.join()
    ctx.synchronize()
    ctx.set_profiling(True)   # HIP events on the launch stream around every kernel (ra_set_profiling)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.join()
    barrier()
    dt = time.perf_counter() - t0
    ctx.synchronize()         # raises on device-side error flags
    kms, calls = ctx.kernel_ms()
    ctx_launches = ctx.kernel_launches()
    ctx.set_profiling(False)
    st = ctx.stats()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out_rms = float(out.double().pow(2).mean().sqrt().item())
    frames_per_step_gpu = S * n
    total_frames = frames_per_step_gpu * args.steps * world
    value = total_frames / dt

    result = None
    if rank == 0:
        launches = ctx_launches
        synth_ms = (kms["propagate"] + kms["render"]) / max(calls, 1)
        # the sine-bank renderer carries ~75 % of the path's VALU work; the analysis launches overlap it and their
        # event-to-event time is mostly waiting for CU slots, so "largest summed duration" would not name the kernel that bounds the step
        dom = "render" if kms["render"] > 0 else max(kms, key=kms.get)
        dom_launches = max(launches[dom], 1)
        dom_ms = kms[dom] / dom_launches                       # average duration of ONE launch (HIP events on its stream)
        launches_per_step = dom_launches / max(calls, 1)
        algo_bytes = ALGO_BYTES_PER_FRAME * frames_per_step_gpu / launches_per_step   # a launch covers 1 / launches_per_step of the block
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        flops = algo_flops_per_frame(N, H, P)
        KNAME = {"analysis": "ra_analysis_kernel", "track": "ra_track_kernel", "vocoder": "ra_voc_*_kernel",
                 "propagate": "ra_synth_propagate_kernel", "render": "ra_synth_render_mono_kernel", "history": "ra_roll_history_kernel"}
        pmc, pmc_src = measured_counters(KNAME[dom], S, n, N, H, P)
        traffic = (2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0 / pmc.get("launches", 1) if pmc else None
        valu = {"algorithmic_flop_per_frame": flops,
                "achieved_tflops_fp64": flops * frames_per_step_gpu / (synth_ms * 1e-3 + 1e-30) / 1e12 if synth_ms > 0 else None,
                "peak_tflops_fp64_vector": FP64_VALU_PEAK_TFLOPS}
        if pmc and "SQ_INSTS_VALU" in pmc:  # issue-slot view of the same kernel: wave-instructions / s against 1 per 4 cycles per SIMD
            ginst = pmc["SQ_INSTS_VALU"] / (kms[dom] / max(calls, 1) * 1e-3) / 1e9
            valu.update({"kernel": KNAME[dom], "valu_insts_per_step": pmc["SQ_INSTS_VALU"], "valu_insts_per_frame_channel":
                         pmc["SQ_INSTS_VALU"] * 64.0 / (frames_per_step_gpu * P), "issue_rate_ginst_s": ginst,
                         "peak_issue_rate_ginst_s": VALU_PEAK_GINST, "issue_frac": ginst / VALU_PEAK_GINST, "counters_from": pmc_src})
        result = {
            "metric": "audio frames/sec (48 kHz, 4096-pt FFT, 64 partials) per GPU; CPU ref x",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "issue": "stream-ordered" if args.stream_ordered else "overlapped blocks (ra_process_device_async)",
            "config": {"workload": f"configs[1]: {S} mono streams/GPU x {args.seconds:g} s @ 48 kHz, window {N} (Gaussian 4 sigma) / FFT {N}, "
                                   f"stride {H} (75% overlap), {P} partials, analysis_volume 1, stereo_spread 0, post none",
                       "streams_per_gpu": S, "frames_per_stream_per_step": n, "sharding": f"streams x{world}, no collective"},
            "x_realtime_per_stream": value / (sr * S * world),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": KNAME[dom],
                         "kernel_ms_per_launch": dom_ms, "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "the path is fp64-VALU issue bound, not HBM bound (SURVEY.md §8d): see valu.issue_frac"},
            "valu": valu,
            "kernel_ms_per_step": {k: v / max(calls, 1) for k, v in kms.items()},
            "kernel_launches_per_step": {k: v / max(calls, 1) for k, v in launches.items()},
            "checks": {"out_rms": out_rms, "dropped_note_on": int(st.dropped_note_on), "error_flags": int(st.error_flags)},
        }

    # CPU baseline: the oracle (a port; the reference itself cannot be built here) on the host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        cores = host_cores()
        cs = args.cpu_streams or S   # the oracle is fast enough to take the whole batch (~10 s of CPU work on 16 cores)
        cn = n
        hv = voice[:cs, :cn].cpu().numpy()
        ocfg = oracle_lib.default_config(sample_rate=sr, window_size=N, stride=H, n_channels=P, analysis_volume=1.0, stereo_spread=0.0,
                                         carrier_volume=0.0, seed=1234)
        oracle_lib.run_offline_batch(ocfg, hv[:1, :sr], n_threads=1)  # page in
        t1 = time.perf_counter()
        ref = oracle_lib.run_offline_batch(ocfg, hv, n_threads=cores)
        cdt = time.perf_counter() - t1
        t1 = time.perf_counter()
        oracle_lib.run_offline_batch(ocfg, hv[:1], n_threads=1)
        cdt1 = time.perf_counter() - t1
        # the same sample through the GPU path, from a fresh context, as a parity spot check at bench size
        with api.Context(cfg, cs, local_rank) as c2:
            got = c2.process_host(hv)
        err = np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2, axis=(1, 2)))
        result["cpu_baseline"] = {"value": cs * cn / cdt, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": f"{cs} of the {S} streams x {cn / sr:g} s, oracle (g++ -O2, double), one std::thread per core",
                                  "single_core_frames_per_s": cn / cdt1, "seconds": cdt,
                                  "gpu_vs_oracle_max_rms": float(err.max())}
        result["gpu_over_cpu"] = value / (cs * cn / cdt)

    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
