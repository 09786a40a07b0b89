#!/usr/bin/env python3
"""Headline benchmark: audio frames/s of the rt.resynth analysis -> resynthesis hot path on MI355X.

One "step" = one pass of the whole hot path (analysis + tracking + sine-bank synthesis) over one batch of
synthetic input: BASELINE.json configs[1] = 256 mono streams x 10 s @ 48 kHz, 4096-pt window, 75 % overlap
(H = 1024), 64 partials.  Consecutive steps continue the same streams (steady state);
inputs are resident in HBM before the timed region.  N > 1: every rank runs its own streams on its own
GPU (streams are independent: no collective on the data path; the barrier and the max-over-ranks of the
timing go over gloo on CPU tensors), value = all ranks' frames / max-over-ranks time.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

The JSON line also carries, outside the timed region of the headline (N = 1 only):
  kernels        per kernel: summed launch time per step while the stages overlap (`overlapped_ms`) and with the chip to
                 itself (`alone_ms`: one extra step with ra_set_pipeline_serial), launches per step, achieved algorithmic GB/s
  roofline       for the kernel with the largest alone_ms (derived, not asserted)
  other_configs  configs[2] (FFT vocoder; band-pass bank fed by the carrier synth), configs[3] and configs[1] with
                 stereo_spread = 1 at BASELINE size: ms_per_step, frames/s, per-kernel ms, error flags
  cpu_baseline   the oracle on the host cores over a bounded sample, with a parity spot check at bench size
"""
import argparse
import hashlib
import json
import math
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
SR = 48000
STAGES = ("analysis", "track", "vocoder", "propagate", "render", "history", "prep")
COMPACT_LIMIT = 4096           # bytes: the driver keeps an 8 KB tail of stdout and parses its last line


def algo_flops_per_frame(N, H, P):
    """FFT 5 N log2 N / H + window/|X|^2/peaks 12 N / H + sine bank 30 P."""
    return 5.0 * N * math.log2(N) / H + 12.0 * N / H + 30.0 * P


def csrc_sha256():
    """Hash of the kernel sources the loaded library was built from: PMC counters are only quoted for the build they were taken on."""
    d = os.path.join(ROOT, "cpp_audio_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".cuh", ".h")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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


def measured_counters(S, n, N, H, P):
    """Per-step PMC sums from the committed rocprofv3 passes (profiles/r*/traffic.json: separate --pmc runs of FETCH_SIZE, WRITE_SIZE
    and SQ_INSTS_VALU) — PMC counters cannot be read from inside this process.  Only a profile of exactly this workload shape taken on
    exactly this build of the kernels (csrc_sha256) counts; anything else returns (None, why)."""
    import glob
    here = csrc_sha256()
    why = "no profiles/r*/traffic.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
            w = doc["workload"]
            if (w["streams"], w["frames_per_stream"], w["window"], w["stride"], w["partials"]) != (S, n, N, H, P):
                why = f"{os.path.relpath(path, ROOT)} is for another workload shape"
                continue
            if doc.get("csrc_sha256") != here:
                why = f"{os.path.relpath(path, ROOT)} was collected on another build (csrc_sha256 {doc.get('csrc_sha256')} != {here})"
                continue
            return doc, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, why


def workload_evidence(name, streams, frames, bytes_per_frame=ALGO_BYTES_PER_FRAME):
    """The committed rocprofv3 evidence of one `other_configs` workload (profiles/r*/<name>/counters.json, scripts/collect_profiles.sh workloads):
    per kernel the average launch duration, launches per step, VALU issue fraction and HBM bytes against the algorithmic bytes, and the
    roofline block of the workload's dominant kernel (largest time per step).  Only a profile of this build (csrc_sha256) and shape counts."""
    import glob
    here = csrc_sha256()
    why = f"no profiles/r*/{name}/counters.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name, "counters.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        rel = os.path.relpath(path, ROOT)
        if doc.get("csrc_sha256") != here:
            why = f"{rel} was collected on another build (csrc_sha256 {doc.get('csrc_sha256')} != {here})"
            continue
        w = doc["workload"]
        if (w["streams"], w["frames_per_stream_per_step"]) != (streams, frames):
            why = f"{rel} is for another workload size"
            continue
        algo_step = bytes_per_frame * streams * frames
        kern = {}
        for k, c in doc["kernels"].items():
            if "total_ms_per_step" not in c:
                continue
            e = {"launches_per_step": c["launches_per_step"], "ms_per_step": c["total_ms_per_step"], "avg_launch_ms": c["avg_launch_ms"]}
            if "SQ_INSTS_VALU" in c:
                e["valu_insts_per_step"] = c["SQ_INSTS_VALU"]
                e["issue_frac"] = c["SQ_INSTS_VALU"] / (c["total_ms_per_step"] * 1e-3) / 1e9 / VALU_PEAK_GINST
            if "FETCH_SIZE_KB" in c and "WRITE_SIZE_KB" in c:
                e["hbm_bytes_per_step"] = (2.0 * c["FETCH_SIZE_KB"] + c["WRITE_SIZE_KB"]) * 1024.0
                e["hbm_over_algorithmic"] = e["hbm_bytes_per_step"] / algo_step
            if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
                e["lds_conflict_frac"] = (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
            kern[k] = e
        if not kern:
            continue
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        d = kern[dom]
        algo_launch = algo_step / d["launches_per_step"]
        achieved = algo_launch / (d["avg_launch_ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": algo_launch, "traffic": d.get("hbm_bytes_per_step", 0.0) / d["launches_per_step"] if "hbm_bytes_per_step" in d else None,
                "kernel_ms_per_launch": d["avg_launch_ms"], "issue_frac": d.get("issue_frac"),
                "note": "fp64-VALU issue bound, not HBM bound: issue_frac = SQ_INSTS_VALU / kernel time / 614 G wave-instructions per second"}
        return {"from": rel, "roofline": roof, "kernels": kern}
    return {"from": None, "why": why}


def synth_inputs_torch(n_streams, n_frames, sr, seed, device):
    """Seeded synthetic streams of the SURVEY.md §8(d) recipe, generated on the GPU (float32, peak 0.5)."""
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


def carrier_saw_torch(S, n, sr, device):
    """configs[2] carrier from a file: 110 Hz saw (1 % detune per stream), amplitude 0.5."""
    import torch
    t = torch.arange(n, dtype=torch.float64, device=device) / sr
    f = 110.0 * (1.0 + 0.01 * torch.arange(S, dtype=torch.float64, device=device)).unsqueeze(1)
    return (0.5 * (2.0 * torch.remainder(f * t, 1.0) - 1.0)).to(torch.float32).contiguous()


def carrier_chords(S, n_steps_total, n):
    """configs[2] carrier from the carrier synth: per stream a four-note chord struck anew every second and released 0.8 s later."""
    rows = []
    for s in range(S):
        keys = (45 + s % 12, 52 + s % 7, 57 + s % 5, 64 + s % 3)
        t = 0
        while t < n_steps_total * n:
            for j, k in enumerate(keys):
                rows.append((t + 97 * j, s, 0, k, 0, 0.0, 0.8))
            for j, k in enumerate(keys):
                rows.append((t + int(0.8 * SR) + 11 * j, s, 1, k, 0, 0.0, 0.0))
            t += SR
    return rows


class Runner:
    """One context on one GPU, its inputs resident in HBM; step() issues one block the way an offline job does."""

    def __init__(self, api, torch, dev, local_rank, cfg_kw, S, n, seed, carrier=None, stream_ordered=False, total_steps=0, setup_block=True):
        self.api, self.torch, self.S, self.n = api, torch, S, n
        self.voice = synth_inputs_torch(S, n, SR, seed, dev)
        self.carrier = carrier_saw_torch(S, n, SR, dev) if carrier == "wav" else None
        self.out = torch.empty(S, n, 2, dtype=torch.float32, device=dev)
        self.cfg = api.default_config(sample_rate=SR, **cfg_kw)
        self.ctx = api.Context(self.cfg, S, local_rank)
        self.ctx.set_hip_stream(torch.cuda.current_stream().cuda_stream)
        self.stream_ordered = stream_ordered
        self.chords = carrier == "synth"
        self.total_steps = total_steps
        # set-up, not a step: a 2 s block makes HIP load the code objects and creates the context's internal streams and events, then
        # the streams are put back to their initial state; the per-block scratch (GBs) is still allocated by the first warm-up step
        # (scripts/profile_workload.py leaves it out: under rocprofv3 every launch of the run should belong to one of its equal steps)
        if setup_block:
            self.ctx.process_device(self.voice.data_ptr(), self.carrier.data_ptr() if self.carrier is not None else 0, min(n, 2 * SR), self.out.data_ptr())
            self.ctx.synchronize()
            self.ctx.reset()
        if self.chords:
            self.ctx.carrier_events(carrier_chords(S, total_steps, n))

    # Steps are consecutive blocks of the same streams, issued like an offline job issues them (ra_process_device_async): the caller's
    # stream is not made to wait between blocks, so block k + 1's analysis and tracking overlap block k's rendering; every block still
    # does all of its work inside the timed region, which ends with a join + device synchronisation.
    def step(self):
        self.ctx.process_device(self.voice.data_ptr(), self.carrier.data_ptr() if self.carrier is not None else 0, self.n, self.out.data_ptr(),
                                overlap=not self.stream_ordered)

    def finish(self):
        self.ctx.join()

    def kernel_split(self, steps):
        """(ms per step, launches per step) per stage over `steps` further steps, HIP events on the launch streams."""
        self.ctx.set_profiling(True)
        for _ in range(steps):
            self.step()
        self.ctx.join()
        self.ctx.synchronize()
        kms, calls = self.ctx.kernel_ms()
        launches = self.ctx.kernel_launches()
        self.ctx.set_profiling(False)
        calls = max(calls, 1)
        return {k: kms[k] / calls for k in STAGES}, {k: launches[k] / calls for k in STAGES}

    def close(self):
        self.ctx.close()
        del self.voice, self.out, self.carrier
        self.torch.cuda.empty_cache()


def run_shim_bench():
    """The reference's real drop-in seam (RecordF + ComputeFunc, simple_audio_context.h:24-26) through the C ABI, driven from compiled C++ the way
    SimpleAudioOutContext::step drives its computes (tests/host/shim_bench.cpp): 1, 64 and 256 streams on one context, 16-frame tiles.  Runs as a
    child process BEFORE this process touches the GPU (a process that has initialised HIP must not fork + exec)."""
    import subprocess
    import tempfile
    lib_dir = os.path.join(ROOT, "cpp_audio_amd")
    exe = os.path.join(tempfile.mkdtemp(prefix="ra_shim_"), "shim_bench")
    try:
        subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "shim_bench.cpp"), "-o", exe, "-L", lib_dir, "-lresynth_amd", f"-Wl,-rpath,{lib_dir}",
                        "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, timeout=300)
        r = subprocess.run([exe, "1", "10", "64", "5", "256", "3"], check=True, capture_output=True, timeout=300, text=True)
        runs = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
        return {"config": "ComputeFunc shim from C++ (tests/host/shim_bench.cpp): per 16-frame tile every stream is fed, then every stream's compute runs; window 4096, "
                          "stride 1024, 64 partials; a synth-only context renders one hop ahead per launch sequence and serves the tiles from a ring",
                "runs": runs, "x_realtime_1_stream": runs[0]["x_realtime"], "x_realtime_64_streams": runs[1]["x_realtime"], "x_realtime_256_streams": runs[2]["x_realtime"]}
    except Exception as e:  # (no compiler on the box, ...)
        return {"config": "ComputeFunc shim from C++ (tests/host/shim_bench.cpp)", "error": repr(e)}


def voice_workload(api, which, S, seconds, blocks):
    """The gen.voice workloads of the bench line: (config, rows, description).  16 channels per synth, a note every 50 ms per synth."""
    # note-offs come after the note has played out (a note-off before that loses the channel for good, soundengine.h:345-353,470-480), and
    # the note-ons are spaced so that the 16 channels suffice: nothing is dropped
    if which == "sweep":
        cfg = api.voice_default_config(seed=99, random_pan=1, length_exponent=2.0)
        desc, every, hold = "SWEEP: a 292 ms sweep (100 Hz -> 20 kHz) every 50 ms", SR // 20, (SR * 4) // 10
    elif which == "birds":
        cfg, name = api.voice_program(api.VOICE_BIRDS, 8, seed=99)       # "Talkative bird": six Markov steps, frequency transitions, pauses: ~4 s of ~10 ramps
        desc, every, hold = f"BIRDS program '{name}': a note-on every 350 ms, released after 5 s", (SR * 35) // 100, 5 * SR
    elif which == "wind":
        # "Light rain in a car": pink noise through a band-pass of 89 + 89 one-pole sections whose centre and width move every ~11 samples.
        # A WIND note is one ramp that sounds until its note-off (which loses the channel for good): four notes per synth, held
        cfg, name = api.voice_program(api.VOICE_WIND, 6, seed=99)
        rows = []
        for s in range(S):
            rows += [(k * (SR // 2) + 7 * (s % 64), s, 0, k, 220.0 + 20.0 * k, 0.5) for k in range(4)]
        return cfg, rows, f"WIND program '{name}' (filters of order 89): four notes per synth, held"
    else:
        cfg, name = api.voice_program(api.VOICE_ROBOTS, 1, seed=99)      # "Communication": 16 + 14 Markov steps: ~2 s of ~20 tones
        desc, every, hold = f"ROBOTS program '{name}': a note-on every 250 ms, released after 3.5 s", SR // 4, (7 * SR) // 2
    rows = []
    for s in range(S):
        per = []
        for k in range(int(blocks * seconds * SR) // every):
            on = k * every + 7 * (s % 64)
            per.append((on, s, 0, k, 220.0 + 20.0 * (k % 23), 0.5))
            per.append((on + hold, s, 1, k, 0.0, 0.0))
        rows += sorted(per, key=lambda r: r[0])
    return cfg, rows, desc


def time_voice(api, torch, dev, which, S, seconds, steps):
    """gen.voice, S synths of 16 channels, output resident in HBM.  Blocks are issued asynchronously on a stream of
    their own (ra_voice_process_device_async): the host's share of block k + 1 (events, Markov chains, schedules, job tables) runs while the
    device renders block k, as the headline's blocks do.  `kernel_ms` = HIP events around the two kernels of a block."""
    n = int(round(seconds * SR))
    cfg, rows, desc = voice_workload(api, which, S, seconds, steps + 4)
    out = torch.empty(S, n, 2, dtype=torch.float32, device=dev)
    st = torch.cuda.Stream()
    with api.Voice(cfg, S, 0) as v:
        v.set_hip_stream(st.cuda_stream)
        v.events(rows)
        for _ in range(3):                    # warm-up blocks: every one of the object's double / triple buffers has been sized once
            v.process_device(n, out.data_ptr())
        v.kernel_ms(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            v.process_device_async(n, out.data_ptr())
        t_host = (time.perf_counter() - t0) / steps
        v.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kms, launches = v.kernel_ms()
        dropped = v.dropped_note_ons()
        rms = float(out.double().pow(2).mean().sqrt().item())
        # the same block twice more, one at a time: what a caller that waits for every block sees
        t0 = time.perf_counter()
        v.process_device(n, out.data_ptr())
        dt_sync = time.perf_counter() - t0
    return {"config": f"gen.voice {desc} on each of {S} synths x 16 channels, {seconds:g} s per step, output resident", "streams": S, "seconds_per_step": seconds,
            "steps": steps, "ms_per_step": dt * 1e3, "ms_per_step_one_block_at_a_time": dt_sync * 1e3, "host_ms_per_step": t_host * 1e3,
            "kernel_ms_per_step": {k: x / max(launches, 1) for k, x in kms.items()}, "frames_per_s": S * n / dt, "x_realtime": seconds / dt,
            "dropped_note_on": int(dropped), "out_rms": rms}


def workloads(N=4096, H=1024, P=64):
    """The single-GPU workloads of the bench line besides the headline, by the name their profiles are kept under (profiles/rNN/<name>/).
    Resynthesis workloads: (description, config, streams, carrier source); gen.voice workloads: (which, synths)."""
    parity = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
    voc = dict(parity, analysis_volume=0.0, voice_volume=0.0, carrier_volume=0.0, vocoder_volume=1.0, vocoder_count_bands=32, vocoder_min_freq=100.0,
               vocoder_max_freq=20000.0, vocoder_modulator_window_size_seconds=0.10, vocoder_stride_seconds=0.005)
    return {
        "configs1": ("configs[1] (the headline shape): 256 mono streams, N 4096, H 1024, P 64", dict(parity, window_size=N, stride=H, n_channels=P), 256, None),
        "north_star": ("north_star operating point: 1000 concurrent 48 kHz mono streams, 4096-pt window, 75 % overlap, 64 partials, on one MI355X (= one rank of configs[4])",
                       dict(parity, window_size=N, stride=H, n_channels=P), 1000, None),
        "stereo": ("configs[1] with stereo_spread = 1 (the reference default): 256 streams, N 4096, H 1024, P 64",
                   dict(parity, window_size=N, stride=H, n_channels=P, stereo_spread=1.0), 256, None),
        "voc_fft": ("configs[2] FFT vocoder: 256 streams, 32 bands, S = 241, modulator FFT 8192, carrier FFT 512, carrier from a file", voc, 256, "wav"),
        "voc_bank": ("configs[2] band-pass bank (vocoder_kind 1): 256 streams, 32 bands, carrier from the carrier synth (sine + saw voices, 16 channels)",
                     dict(voc, vocoder_kind=1, carrier_n_channels=16, vocoder_carrier_square_volume=0.0, vocoder_carrier_sine_volume=0.7, vocoder_carrier_saw_volume=0.3),
                     256, "synth"),
        "configs3": ("configs[3]: 1000 stereo = 2000 mono streams, N = 8192, H = 2048, P = 128, autotune major / C",
                     dict(parity, window_size=8192, stride=2048, n_channels=128, use_autotune=1, autotune_type=0, autotune_scale_mode=0, autotune_root_note=0), 2000, None),
        "voice_sweep": ("sweep", 256), "voice_birds": ("birds", 256), "voice_robots": ("robots", 256), "voice_wind": ("wind", 256),
    }


def time_config(api, torch, dev, name, cfg_kw, S, seconds, steps, carrier=None):
    """An `other_configs` entry: full-size timing of another BASELINE config (same issue mode and clock as the headline)."""
    n = int(round(seconds * SR))
    warm = 3
    r = Runner(api, torch, dev, 0, cfg_kw, S, n, 4321, carrier=carrier, total_steps=warm + steps + 7)
    for _ in range(warm):
        r.step()
    r.finish()
    r.ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.step()
    r.finish()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r.ctx.synchronize()
    over, launches = r.kernel_split(2)
    r.ctx.set_pipeline_serial(True)
    alone, _ = r.kernel_split(3)   # (three steps: a single serialised step is a noisy sample of the two chain kernels)
    r.ctx.set_pipeline_serial(False)
    st = r.ctx.stats()
    res = {"config": name, "streams": S, "seconds_per_step": seconds, "steps": steps, "ms_per_step": dt * 1e3, "frames_per_s": S * n / dt,
           "x_realtime_per_stream": n / SR / dt,
           "kernel_ms_per_step": {"overlapped": {k: round(v, 3) for k, v in over.items()}, "alone": {k: round(v, 3) for k, v in alone.items()}},
           "kernel_launches_per_step": launches, "error_flags": int(st.error_flags), "dropped_note_on": int(st.dropped_note_on),
           "carrier_dropped_note_on": int(st.carrier_dropped_note_on), "out_rms": float(r.out.double().pow(2).mean().sqrt().item())}
    r.close()
    return res


def _sig(x, digits=5):
    """floats rounded to `digits` significant figures (the compact line carries figures, not bit patterns)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(result, details_path=None):
    """The LAST stdout line: the headline object the driver parses, < COMPACT_LIMIT bytes whatever the detail holds (the single
    30 KB line of r03 did not survive the driver's 8 KB tail).  Everything else goes to the detail file / the earlier stdout line."""
    c = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                       "x_realtime_per_stream", "gpu_over_cpu", "csrc_sha256"))
    cfg = result.get("config", {})
    c["config"] = _pick(cfg, ("workload", "streams_per_gpu", "streams_total", "frames_per_stream_per_step", "sharding"))
    roof_keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch", "launches_per_step", "algorithmic_bytes_per_launch",
                 "issue_frac", "kernel_selected_by")
    if "roofline" in result:
        c["roofline"] = _pick(result["roofline"], roof_keys)
    if "roofline_second" in result:
        c["roofline_second"] = _pick(result["roofline_second"], roof_keys[:-1])
    if "cpu_baseline" in result:
        c["cpu_baseline"] = _pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "gpu_vs_oracle_max_rms"))
    v = result.get("valu", {})
    c["valu"] = _pick(v, ("step_issue_frac", "achieved_tflops_fp64", "effective_clock_ghz", "step_issue_frac_at_held_clock"))
    k = result.get("kernels", {})
    c["kernel_ms_per_step"] = {name: {"alone": e.get("alone_ms_per_step"), "overlapped": e.get("overlapped_ms_per_step")} for name, e in k.items() if name != "history"}
    others = {}
    for o in result.get("other_configs", []) or []:
        if "workload" in o and "ms_per_step" in o:
            others[o["workload"]] = o["ms_per_step"]
            if o["workload"] == "north_star":
                c["north_star"] = {"ms_per_step": o["ms_per_step"], "x_realtime": o.get("x_realtime_per_stream"), "streams": o.get("streams"),
                                   "pcie_inclusive_ms_per_step": (o.get("pcie_inclusive") or {}).get("ms_per_step"),
                                   "pcie_inclusive_pinned_ms_per_step": (o.get("pcie_inclusive") or {}).get("pinned_ms_per_step"),
                                   "pcie_inclusive_pipelined_ms_per_step": (o.get("pcie_inclusive") or {}).get("pipelined_ms_per_step")}
        elif "error" in o:
            others["error"] = str(o["error"])[:120]
    if others:
        c["other_ms_per_step"] = others
    if "pcie_inclusive" in result:
        c["pcie_inclusive"] = _pick(result["pcie_inclusive"], ("ms_per_step", "pinned_ms_per_step", "pipelined_ms_per_step", "streams", "GBs_host_link", "pinned_GBs_host_link", "pipelined_GBs_host_link"))
    if result.get("retimed"):
        c["retimed"] = _pick(result["retimed"], ("first_ms_per_step", "busiest_queue_ms_per_step"))
    c["checks"] = result.get("checks")
    if details_path:
        c["details"] = details_path
    # value and ms_per_step keep every digit (the driver checks one against the other); the diagnostic blocks carry 5 significant figures
    c = {k: (v if k in ("value", "ms_per_step") else _sig(v)) for k, v in c.items()}
    line = json.dumps(c, separators=(",", ":"))
    # never let an unexpected field push the line past the limit: drop the optional blocks, least important first
    for drop in ("other_ms_per_step", "kernel_ms_per_step", "roofline_second", "checks", "retimed", "pcie_inclusive", "valu", "north_star"):
        if len(line) < COMPACT_LIMIT:
            break
        c.pop(drop, None)
        line = json.dumps(c, separators=(",", ":"))
    return line


def emit(result, details_path, out=None):
    """Writes the full result to `details_path` (and as an EARLIER stdout line), then the compact headline as the LAST stdout line."""
    out = out or sys.stdout
    where = None
    if details_path:
        try:
            with open(details_path, "w") as f:
                json.dump(result, f, indent=1)
            where = os.path.relpath(details_path, ROOT) if os.path.abspath(details_path).startswith(ROOT) else details_path
        except OSError:
            where = None
    out.write(json.dumps({"bench_details": result}) + "\n")
    line = compact_line(result, where)
    out.write(line + "\n")
    out.flush()
    return line


def pcie_inclusive(api, torch, cfg, S, n, local_rank):
    """The PCIe-inclusive rate DESIGN.md quotes beside `value` (never value itself): one block of S streams through ra_process_host — the block's
    input up, the block, float32 stereo down, synchronised — on a fresh context (its third block: the first two size the staging), from pageable
    memory (numpy) and from page-locked memory (ra_host_alloc)."""
    import numpy as np
    hv = (np.random.default_rng(7).standard_normal((S, n), dtype=np.float32) * 0.05)
    res = {"streams": S, "frames_per_stream": n,
           "how": "ra_process_host: H2D of [S][n] float32, the block, D2H of [S][n][2] float32, synchronised; ms_per_step from pageable host memory, "
                  "pinned_ms_per_step from ra_host_alloc memory"}
    with api.Context(cfg, S, local_rank) as c:
        c.process_host(hv[:, :min(n, SR)])
        c.reset()
        c.process_host(hv)
        t0 = time.perf_counter()
        c.process_host(hv)
        dt = time.perf_counter() - t0
        res.update({"ms_per_step": dt * 1e3, "GBs_host_link": 12.0 * S * n / dt / 1e9})
        try:
            with api.PinnedArray((S, n), np.float32) as pin_in, api.PinnedArray((S, n, 2), np.float32) as pin_out:
                pin_in.array[...] = hv
                c.process_host(pin_in.array, out=pin_out.array)
                t0 = time.perf_counter()
                c.process_host(pin_in.array, out=pin_out.array)
                dtp = time.perf_counter() - t0
                res.update({"pinned_ms_per_step": dtp * 1e3, "pinned_GBs_host_link": 12.0 * S * n / dtp / 1e9})
                # ... and with the copies of consecutive one-second chunks overlapping the kernels (ra_process_host_pipelined)
                c.process_host(pin_in.array, out=pin_out.array, chunk_frames=SR)
                t0 = time.perf_counter()
                c.process_host(pin_in.array, out=pin_out.array, chunk_frames=SR)
                dtq = time.perf_counter() - t0
                res.update({"pipelined_ms_per_step": dtq * 1e3, "pipelined_GBs_host_link": 12.0 * S * n / dtq / 1e9, "pipelined_chunk_frames": SR})
        except Exception as e:
            res["pinned_error"] = repr(e)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: configs[1]'s 256; with --gpus 8 and no stream option: configs[4], see below)")
    ap.add_argument("--streams-total", type=int, default=0, help="total streams, sharded over the ranks; overrides --streams.  `--gpus 8` without a stream option "
                    "runs BASELINE configs[4]: 8000 streams over the 8 GPUs = 1000 per rank")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds per stream per step")
    ap.add_argument("--window", type=int, default=4096)
    ap.add_argument("--stride", type=int, default=1024)
    ap.add_argument("--partials", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="only the headline (profiling runs)")
    ap.add_argument("--stream-ordered", action="store_true", help="issue every block with ra_process_device (no overlap between consecutive blocks)")
    ap.add_argument("--cpu-streams", type=int, default=0, help="streams in the CPU baseline sample (0 = all)")
    ap.add_argument("--no-alone", action="store_true", help="skip the serialised extra steps (kernels{}.alone_ms): every ra_* launch of the run then belongs to the "
                    "pipeline, which is what the rocprofv3 --stats summary of profiles/rNN/kernel_stats.csv is taken on")
    ap.add_argument("--clock-stamps", action="store_true", help="diagnostic (needs RESYNTH_AMD_LIB = the `make clk` build): after the timed region, >= 2 s of back-to-back "
                    "steps with s_memtime / s_memrealtime stamps around every analysis / render workgroup; reports the clock the chip held (valu.effective_clock_ghz_measured)")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"), help="file the full result is written to (the last stdout line is the compact headline)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # child processes (the C++ shim harness, the oracle's build) run here, before anything in this process initialises the GPU
    shim_result = None
    if rank == 0 and world == 1:
        if not args.no_other_configs:
            shim_result = run_shim_bench()
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            oracle_lib.lib()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    local_rank %= max(1, torch.cuda.device_count())   # (more ranks than GPUs only happens in the two-rank test on a one-GPU box)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # the data path has no collective (north_star: "no RCCL needed"): the only exchanges are the barriers around the timed region
        # and the max of one scalar, on CPU tensors
        dist.init_process_group("gloo")

    from cpp_audio_amd import api, sharding

    configs4 = args.gpus == 8 and not args.streams and not args.streams_total   # BASELINE configs[4]: "8000 independent streams sharded per GPU"
    if configs4:
        args.streams_total = 8000
    if args.streams_total:
        lo, hi = sharding.shard_streams(args.streams_total, world, rank)
        S, first_stream = hi - lo, lo
    else:
        per = args.streams or 256
        S, first_stream = per, rank * per
    n = int(round(args.seconds * SR))
    N, H, P = args.window, args.stride, args.partials
    head_kw = dict(window_size=N, stride=H, n_channels=P, analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234 + first_stream)
    if os.environ.get("RA_BENCH_MAXR"):   # tuning aid: the per-frame list capacity (LDS of an analysis workgroup)
        head_kw["max_pitches_per_frame"] = int(os.environ["RA_BENCH_MAXR"])
    run = Runner(api, torch, dev, local_rank, head_kw, S, n, 1234 + 100003 * rank, stream_ordered=args.stream_ordered)
    ctx, voice, out = run.ctx, run.voice, run.out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run.step()
    run.finish()
    ctx.synchronize()

    def timed_region():
        ctx.set_profiling(True)   # HIP events on the launch stream around every kernel (ra_set_profiling)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run.step()
        run.finish()
        barrier()
        dt_ = time.perf_counter() - t0
        ctx.synchronize()         # raises on device-side error flags
        kms_, calls_ = ctx.kernel_ms()
        launches_ = ctx.kernel_launches()
        ctx.set_profiling(False)
        return dt_, kms_, calls_, launches_

    dt, kms, calls, launches = timed_region()
    # A stall that is not the GPU's work (seen once in ~ a hundred runs on the shared boxes: 240 ms of wall clock around five steps whose queues
    # were busy for 10 ms each) is not a measurement of this path: if the K steps took more than twice what their busiest queue was busy
    # for (HIP events), they are timed once more — K steps again, the same way — and the line says so (`retimed`).
    retimed = None
    busiest = max(kms[k] for k in STAGES) / max(calls, 1) * 1e-3
    flag = 1.0 if (busiest > 0 and dt / args.steps > 2.0 * busiest + 1e-3) else 0.0
    if world > 1:
        ft = torch.tensor([flag], dtype=torch.float64)
        dist.all_reduce(ft, op=dist.ReduceOp.MAX)
        flag = float(ft[0].item())
    if flag:
        retimed = {"first_ms_per_step": dt / args.steps * 1e3, "busiest_queue_ms_per_step": busiest * 1e3,
                   "why": "the first K steps took more than twice the time their busiest queue was busy (a stall outside the GPU's work); K steps were timed again, once"}
        dt, kms, calls, launches = timed_region()
    st = ctx.stats()
    total_streams = S
    if world > 1:
        tt = torch.tensor([dt, float(S)], dtype=torch.float64)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt, total_streams = float(mx[0].item()), int(round(tt[1].item()))

    out_rms = float(out.double().pow(2).mean().sqrt().item())
    frames_per_step_gpu = S * n
    value = total_streams * n * args.steps / dt
    ms_per_step = dt / args.steps * 1e3

    clock = None
    if rank == 0 and args.clock_stamps:
        # MI355X_MICROARCH.md, DVFS give-back item 6: the in-kernel clock after >= 2 s of back-to-back launches on the real (random) data
        heat = int(math.ceil(2.0 / (ms_per_step * 1e-3)))
        ctx.set_clock_stamps(True)
        for _ in range(heat):
            run.step()
        run.finish()
        ctx.synchronize()
        ctx.set_clock_stamps(True)   # clears the sums: what follows is measured on a chip that has been busy for 2 s
        m_steps = max(20, heat // 4)
        t1 = time.perf_counter()
        for _ in range(m_steps):
            run.step()
        run.finish()
        ctx.synchronize()
        dt_clk = (time.perf_counter() - t1) / m_steps
        cs = ctx.clock_stamps()
        ctx.set_clock_stamps(False)
        clock = {"analysis_ghz": cs["analysis"][0], "render_ghz": cs["render"][0], "analysis_workgroups": cs["analysis"][1], "render_workgroups": cs["render"][1],
                 "steps_before": heat, "steps_measured": m_steps, "ms_per_step_while_stamping": dt_clk * 1e3, "csrc_sha256": csrc_sha256(),
                 "how": "diagnostic build (-DRA_CLOCK_STAMPS=1): lane 0 of every analysis / render workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its "
                        "start and end; GHz = 0.1 x sum of shader ticks / sum of 100 MHz ticks over the measured steps (duration-weighted mean over workgroups)"}

    result = None
    if rank == 0:
        calls = max(calls, 1)
        over_ms = {k: kms[k] / calls for k in STAGES}
        lps = {k: launches[k] / calls for k in STAGES}
        # every kernel with the chip to itself: three more steps, their stages serialised on one stream (outside the timed region)
        if args.no_alone:
            alone_ms = {k: None for k in STAGES}
        else:
            ctx.set_pipeline_serial(True)
            alone_ms, _ = run.kernel_split(3)
            ctx.set_pipeline_serial(False)
        mono = head_kw["stereo_spread"] == 0.0
        KNAME = {"analysis": "ra_analysis_kernel", "track": "ra_track_kernel", "vocoder": "ra_voc_*_kernel", "propagate": "ra_synth_propagate_kernel",
                 "render": "ra_synth_render_mono_kernel" if mono else "ra_synth_render_kernel", "history": "ra_roll_history_kernel", "prep": "ra_track_prep_kernel"}
        flops = algo_flops_per_frame(N, H, P)
        pmc_doc, pmc_src = measured_counters(S, n, N, H, P)
        kernels = {}
        for k in STAGES:
            if lps[k] == 0:
                continue
            per_launch_alone = alone_ms[k] / lps[k] if alone_ms[k] is not None else None
            per_launch_over = over_ms[k] / lps[k]
            algo_bytes = ALGO_BYTES_PER_FRAME * frames_per_step_gpu / lps[k]   # a launch covers 1 / launches_per_step of the block
            e = {"kernel": KNAME[k], "launches_per_step": lps[k], "alone_ms_per_step": alone_ms[k], "overlapped_ms_per_step": over_ms[k],
                 "alone_ms_per_launch": per_launch_alone, "overlapped_ms_per_launch": per_launch_over,
                 "algorithmic_GBs_alone": algo_bytes / (per_launch_alone * 1e-3) / 1e9 if per_launch_alone else None}
            c = pmc_doc["kernels"].get(KNAME[k]) if pmc_doc else None
            if c:
                e["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE_KB"] + c["WRITE_SIZE_KB"]) * 1024.0 / c.get("launches", 1)
                if "SQ_INSTS_VALU" in c:
                    e["valu_insts_per_step"] = c["SQ_INSTS_VALU"]
                    if alone_ms[k]:
                        e["issue_frac_alone"] = c["SQ_INSTS_VALU"] / (alone_ms[k] * 1e-3) / 1e9 / VALU_PEAK_GINST
            kernels[k] = e
        # The dominant kernel is derived, by the rule the rocprofv3 summary is read with: the kernel with the largest SUMMED launch
        # duration over the timed region — HIP events around every launch, on the stream it is launched on; since r04 the tracking pre-pass is
        # timed as a kernel of its own instead of inside the analysis stage.  The runner-up (analysis or render, whichever is not dominant) is
        # reported beside it as `roofline_second`.  Launch durations inside the pipeline overlap each other (co-resident kernels stretch each
        # other); `kernels{}.alone_ms_per_step` is each kernel with the chip to itself.
        ranked = sorted((k for k in kernels if k != "history"), key=lambda k: -kernels[k]["overlapped_ms_per_step"])

        def roof(k):
            d = kernels[k]
            ms = d["overlapped_ms_per_launch"]          # HIP events over the timed region, on the stream the kernel is launched on
            ab = ALGO_BYTES_PER_FRAME * frames_per_step_gpu / d["launches_per_step"]
            ach = ab / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            r_ = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": d.get("hbm_bytes_per_launch"),
                  "kernel": d["kernel"], "kernel_ms_per_launch": ms, "launches_per_step": d["launches_per_step"], "algorithmic_bytes_per_launch": ab,
                  "summed_launch_ms_per_step": d["overlapped_ms_per_step"]}
            if "valu_insts_per_step" in d and d["overlapped_ms_per_step"] > 0:
                # VALU wave-instructions per second of the kernel's own (overlapped) launch time against 614 G/s; alone: kernels{}.issue_frac_alone
                r_["issue_frac"] = d.get("issue_frac_alone", d["valu_insts_per_step"] / (d["overlapped_ms_per_step"] * 1e-3) / 1e9 / VALU_PEAK_GINST)
            return r_

        total_valu = sum(c.get("SQ_INSTS_VALU", 0.0) for c in pmc_doc["kernels"].values()) if pmc_doc else None
        valu = {"algorithmic_flop_per_frame": flops,
                "achieved_tflops_fp64": flops * frames_per_step_gpu / (ms_per_step * 1e-3) / 1e12,   # whole-step flops / step time
                "frac_of_peak_fp64_vector": flops * frames_per_step_gpu / (ms_per_step * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "peak_tflops_fp64_vector": FP64_VALU_PEAK_TFLOPS, "peak_issue_rate_ginst_s": VALU_PEAK_GINST, "nominal_clock_ghz": 2.4}
        if total_valu:
            valu.update({"valu_insts_per_step_all_kernels": total_valu, "step_issue_frac": total_valu / (ms_per_step * 1e-3) / 1e9 / VALU_PEAK_GINST,
                         "render_valu_insts_per_channel_sample": pmc_doc["kernels"].get(KNAME["render"], {}).get("SQ_INSTS_VALU", 0.0) * 64.0 / (frames_per_step_gpu * P),
                         "counters_from": pmc_src})
            # the clock the chip HELD inside the two throughput kernels (s_memtime / s_memrealtime stamps of the diagnostic build, merged into
            # profiles/rNN/traffic.json by scripts/pmc_aggregate.py): the issue fraction against the nominal 2.4 GHz and against the held clock
            clk = pmc_doc.get("effective_clock_ghz")
            if clk:
                held = min(clk["analysis_ghz"], clk["render_ghz"]) if isinstance(clk, dict) else float(clk)
                valu["effective_clock_ghz"] = held
                valu["effective_clock"] = clk
                valu["step_issue_frac_at_held_clock"] = valu["step_issue_frac"] * 2.4 / held
        else:
            valu["counters"] = f"null: {pmc_src}"
        result = {
            "metric": "audio frames/sec (48 kHz, 4096-pt FFT, 64 partials) per GPU; CPU ref x",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            # configs[4] fixes 1000 streams per GPU: the per-GPU work is what the config names, whatever N would be
            "scaling": "weak" if (not args.streams_total or configs4) else "strong", "vs_baseline": None, "retimed": retimed,
            "dtype": "f64", "data": "synthetic", "issue": "stream-ordered" if args.stream_ordered else "overlapped blocks (ra_process_device_async)",
            "config": {"workload": f"{'configs[4]: 8000 streams over 8 GPUs, ' if configs4 else 'configs[1]: '}{S} mono streams/GPU x {args.seconds:g} s @ 48 kHz, window {N} (Gaussian 4 sigma) / FFT {N}, "
                                   f"stride {H} (75% overlap), {P} partials, analysis_volume 1, stereo_spread 0, post none",
                       "streams_per_gpu": S, "streams_total": total_streams, "frames_per_stream_per_step": n, "sharding": f"streams x{world}, no collective"},
            "x_realtime_per_stream": value / (SR * total_streams),
            "roofline": dict(roof(ranked[0]),
                             kernel_selected_by="largest summed launch duration over the timed region (HIP events per launch on its own stream)",
                             traffic_source=pmc_src if pmc_doc else f"null: {pmc_src}",
                             note="the path is fp64-VALU issue bound, not HBM bound (SURVEY.md §8d): see valu and kernels{}.issue_frac_alone"),
            "valu": valu,
            "kernels": kernels,
            "csrc_sha256": csrc_sha256(),
            "checks": {"out_rms": out_rms, "dropped_note_on": int(st.dropped_note_on), "error_flags": int(st.error_flags)},
        }
        if len(ranked) > 1:
            result["roofline_second"] = roof(ranked[1])
        if clock:
            result["valu"]["effective_clock_ghz_measured"] = clock
        if world > 1 and S >= 1000:   # configs[4] / a north_star rank: what feeding one rank's streams from the host costs (never `value`)
            try:
                result["pcie_inclusive"] = pcie_inclusive(api, torch, run.cfg, S, n, local_rank)
            except Exception as e:
                result["pcie_inclusive"] = {"error": repr(e)}

    # CPU baseline: the oracle (a port; the reference itself cannot be built here) on the host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        cores = host_cores()
        cs = args.cpu_streams or S   # the oracle is fast enough to take the whole batch (~10 s of CPU work on 16 cores)
        cn = n
        hv = voice[:cs, :cn].cpu().numpy()
        okw = {k: v for k, v in head_kw.items()}
        ocfg = oracle_lib.default_config(sample_rate=SR, **okw)
        oracle_lib.run_offline_batch(ocfg, hv[:1, :SR], n_threads=1)  # page in
        t1 = time.perf_counter()
        ref = oracle_lib.run_offline_batch(ocfg, hv, n_threads=cores)
        cdt = time.perf_counter() - t1
        t1 = time.perf_counter()
        oracle_lib.run_offline_batch(ocfg, hv[:1], n_threads=1)
        cdt1 = time.perf_counter() - t1
        # the same sample through the GPU path, from a fresh context, as a parity spot check at bench size
        with api.Context(run.cfg, cs, local_rank) as c2:
            got = c2.process_host(hv)
        err = np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2, axis=(1, 2)))
        result["cpu_baseline"] = {"value": cs * cn / cdt, "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": f"{cs} of the {S} streams x {cn / SR:g} s, oracle (g++ -O2, double), one std::thread per core",
                                  "single_core_frames_per_s": cn / cdt1, "seconds": cdt,
                                  "gpu_vs_oracle_max_rms": float(err.max())}
        result["gpu_over_cpu"] = value / (cs * cn / cdt)
        del ref, got, hv

    run.close()
    # the other BASELINE configs at full size, under the same clock (N = 1 only; not part of `value`)
    if rank == 0 and world == 1 and not args.no_other_configs:
        parity = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
        voc = dict(parity, analysis_volume=0.0, voice_volume=0.0, carrier_volume=0.0, vocoder_volume=1.0, vocoder_count_bands=32, vocoder_min_freq=100.0,
                   vocoder_max_freq=20000.0, vocoder_modulator_window_size_seconds=0.10, vocoder_stride_seconds=0.005)
        others = []
        try:
            wl = workloads(N, H, P)
            for name, steps in (("north_star", 5), ("stereo", 6), ("voc_fft", 5), ("voc_bank", 8), ("configs3", 4)):
                desc, kw, S2, carrier = wl[name]
                o = time_config(api, torch, dev, desc, kw, S2, 10.0, steps, carrier=carrier)
                o["workload"] = name
                if name == "north_star":
                    o["target"] = ">= 10 x real time for each of the 1000 streams (BASELINE.json north_star)"
                    o["pcie_inclusive"] = pcie_inclusive(api, torch, api.default_config(sample_rate=SR, **kw), S2, 10 * SR, local_rank)
                others.append(o)
            for name in ("voice_sweep", "voice_birds", "voice_robots", "voice_wind"):
                o = time_voice(api, torch, dev, wl[name][0], wl[name][1], 10.0, 6)
                o["workload"] = name
                others.append(o)
            if shim_result is not None:
                others.append(shim_result)
        except Exception as e:  # a failure here must not cost the headline line
            others.append({"error": repr(e)})
        # every entry's numbers with their committed evidence: rocprofv3 stats and counters of the same workload on the same build
        for o in others:
            if "workload" in o:
                o["profile"] = workload_evidence(o["workload"], o["streams"], int(round(o["seconds_per_step"] * SR)), 8.0 if o["workload"].startswith("voice_") else ALGO_BYTES_PER_FRAME)
        result["other_configs"] = others

    if rank == 0:
        emit(result, args.details)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
