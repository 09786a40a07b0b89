#!/usr/bin/env python3
"""Tuning aid (GPU box): gen.voice WIND step time for S synths x 4 held notes of one program:  python scripts/time_wind.py <program> <synths> [seconds]
(RA_WIND_SECTIONS_PER_LANE=1|2 forces the filter chains' layout)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cpp_audio_amd import api
prog, S = int(sys.argv[1]), int(sys.argv[2])
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
SR = 48000
n = int(seconds * SR)
cfg, name = api.voice_program(api.VOICE_WIND, prog, seed=99)
rows = []
for s in range(S):
    rows += [(k * (SR // 2) + 7 * (s % 64), s, 0, k, 220.0 + 20.0 * k, 0.5) for k in range(4)]
out = torch.empty(S, n, 2, dtype=torch.float32, device="cuda")
with api.Voice(cfg, S, 0) as v:
    v.events(rows)
    v.process_device(n, out.data_ptr())
    t0 = time.perf_counter()
    for _ in range(3):
        v.process_device_async(n, out.data_ptr())
    v.synchronize()
    dt = (time.perf_counter() - t0) / 3
print(f"{name!r} order {cfg.order_filters} synths {S} layout {os.environ.get('RA_WIND_SECTIONS_PER_LANE', 'auto')}: {dt * 1e3:.2f} ms per {seconds:g} s step")
