#!/usr/bin/env python3
"""Randomised sweep of the output-feedback path (analysis input = voice + gain * delayed mono output, lib.cpp:1626-1652) on the GPU box:
mild loops only (strong ones are chaotic for any two implementations), random delays / strides / block cuts;
audio against the restatement within 1e-4 RMS, whole block and cut block alike."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api, signals

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
REF_ORDER = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 1: the sample-serial renderer (reference_order)
bad = 0
worst = 0.0
for ci in range(n_cfg):
    N = int(rng.choice([1024, 2048, 4096]))
    kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=int(rng.integers(1, 1 << 30)), window_size=N,
              stride=int(rng.choice([N // 4, N // 2, 300])), n_channels=int(rng.choice([16, 64])),
              analysis_output_feedback_gain=float(rng.choice([-0.15, -0.05, 0.05, 0.1, 0.15])),
              output_delay_seconds=float(rng.choice([0.0, 0.0031, 0.0123, 0.05, 0.2, 0.7])))
    n = int(rng.integers(15000, 70000))
    x = signals.make_streams(2, n, base_seed=int(rng.integers(0, 10000)))
    msg = []
    with api.Context(api.default_config(**dict(kw, reference_order=REF_ORDER)), 2, 0) as ctx:
        whole = ctx.process_host(x)
        ctx.reset()
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(1, n, 3)]))
        parts = np.concatenate([ctx.process_host(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
        if ctx.stats().error_flags: msg.append("error flags")
    for s in range(2):
        ref, _, _ = oracle.run_offline(oracle.default_config(**dict(kw, seed=kw["seed"] + s)), x[s])
        r = ref.astype(np.float64)
        e1 = float(np.sqrt(np.mean((whole[s] - r) ** 2))); e2 = float(np.sqrt(np.mean((parts[s] - r) ** 2)))
        worst = max(worst, e1, e2)
        if not (e1 <= 1e-4 and e2 <= 1e-4): msg.append(f"s{s}: rms whole {e1:.2g} cut {e2:.2g}")
    if msg:
        bad += 1
        print(f"config {ci} FAILED: {msg}\n   {kw} n={n} cuts={cuts}")
print(f"{n_cfg - bad} of {n_cfg} random feedback configurations agree with the restatement (worst rms {worst:.2g})")
sys.exit(1 if bad else 0)
