#!/usr/bin/env python3
"""Times the f2 kernels at configs[2] size (256 streams x 10 s): band-pass bank fed by the carrier synth.  Run under
rocprofv3 --kernel-trace --stats for the per-kernel split:  python scripts/time_f2.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cpp_audio_amd import api

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
voc = dict(analysis_volume=0.0, stereo_spread=0.0, seed=1234, voice_volume=0.0, carrier_volume=0.0, vocoder_volume=1.0, vocoder_count_bands=32,
           vocoder_min_freq=100.0, vocoder_max_freq=20000.0, vocoder_kind=1, carrier_n_channels=16, vocoder_carrier_square_volume=0.0,
           vocoder_carrier_sine_volume=0.7, vocoder_carrier_saw_volume=0.3)
r = bench.Runner(api, torch, dev, 0, voc, 256, 480000, 4321, carrier="synth", total_steps=steps + 2)
r.step(); r.finish(); r.ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    r.step()
r.finish(); r.ctx.synchronize()
print("ms per step", (time.perf_counter() - t0) / steps * 1e3)
r.close()
