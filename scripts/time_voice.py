#!/usr/bin/env python3
"""Times the gen.voice SWEEP slice at bench size (the `other_configs` entry) on its own; run under rocprofv3 --kernel-trace --stats for the
kernel / host split:  python scripts/time_voice.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cpp_audio_amd import api
print(json.dumps(bench.time_voice(api, torch, torch.device("cuda", 0), 256, 10.0, 2)))
