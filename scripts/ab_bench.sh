#!/bin/bash
# Tuning aid: runs bench.py (headline only, no CPU baseline) once per library variant given as argument and prints step time + per-kernel ms.
for v in "$@"; do
  RESYNTH_AMD_LIB=$PWD/ab_libs/libresynth_$v.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/ab_$v.log 2>&1
  python3 - gpurun_out/ab_$v.log $v <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), {k: (round(v["alone_ms_per_step"], 2), round(v["overlapped_ms_per_step"], 2)) for k, v in d["kernels"].items() if k != "history"})
PY
done
