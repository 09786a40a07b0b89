"""Tuning aid: one workload's step time for the library named by RESYNTH_AMD_LIB (A/B runs on one box):  python scripts/ab_config.py configs3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cpp_audio_amd import api
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "configs3"
desc, kw, S, carrier = bench.workloads()[name]
o = bench.time_config(api, torch, dev, desc, kw, S, 10.0, 3, carrier=carrier)
print(os.environ.get("RESYNTH_AMD_LIB", "in-tree"), name, round(o["ms_per_step"], 2), o["kernel_ms_per_step"]["alone"])
