#!/bin/bash
# Tuning aid: runs bench.py (no CPU baseline) once per library variant given as argument and prints step time + per-kernel ms.
for v in "$@"; do
  RESYNTH_AMD_LIB=$PWD/ab_libs/libresynth_$v.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
  echo "$v $(grep -o '"ms_per_step[^,]*' gpurun_out/ab_$v.log) $(grep -o '"kernel_ms_per_step[^}]*}' gpurun_out/ab_$v.log)"
done
