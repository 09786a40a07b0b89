"""Tuning aid: times the gen.voice workloads of bench.py (SWEEP / BIRDS / ROBOTS) and splits the step into host time, pre-pass and render."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cpp_audio_amd import api
import bench

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["sweep", "birds", "robots"]
    dev = torch.device("cuda:0")
    for w in which:
        r = bench.time_voice(api, torch, dev, w, S, 10.0, steps)
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
