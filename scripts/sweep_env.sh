#!/bin/bash
# Tuning aid (GPU box): bench.py headline once per "VAR=value[,VAR=value...]" argument, prints step time and per-kernel ms.
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' ')
  env $envs python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/sweep.log 2>&1
  python3 - gpurun_out/sweep.log "$spec" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), {k: (round(v["alone_ms_per_step"], 2), round(v["overlapped_ms_per_step"], 2)) for k, v in d["kernels"].items() if k != "history"})
PY
done
