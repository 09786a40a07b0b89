#!/usr/bin/env python3
"""Randomised sweep of driver (A) (ra_resynth_notes against the restatement of include/resynth.hpp:4-100) on the GPU box."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for ci in range(n_cfg):
    sr = int(rng.choice([22050, 44100, 48000]))
    stride = int(rng.choice([64, 256, 400, 1000]))
    n_notes = int(rng.choice([0, 1, 2, 17, 90, 400]))
    start = np.sort(rng.integers(0, 200, n_notes)).astype(np.int32)
    end = (start + rng.integers(0, 60, n_notes)).astype(np.int32)
    freq = np.exp(rng.uniform(np.log(20.0), np.log(0.49 * sr), n_notes))
    amp = rng.uniform(-60.0, 0.0, n_notes)
    perm = rng.permutation(n_notes)
    start, end, freq, amp = start[perm], end[perm], freq[perm], amp[perm]
    ref = oracle.resynth_notes(start, end, freq, amp, stride, sr)
    got = api.resynth_notes(start, end, freq, amp, stride, sr)
    ok = len(ref) == len(got) and (len(ref) == 0 or float(np.sqrt(np.mean((ref - got) ** 2))) <= 1e-5)
    if not ok:
        bad += 1
        print(f"config {ci} FAILED: sr {sr} stride {stride} notes {n_notes} len {len(ref)}/{len(got)} rms {float(np.sqrt(np.mean((ref[:len(got)] - got[:len(ref)]) ** 2))) if len(ref) and len(got) else None}")
print(f"{n_cfg - bad} of {n_cfg} random note sets agree with the restatement")
sys.exit(1 if bad else 0)
