#!/bin/bash
# Tuning aid (GPU box): per-kernel SQ counters of an un-pipelined step (RA_PIPELINE_CHUNKS=1), one rocprofv3 run per counter group.
# PMC_SCRIPT / PMC_ARGS choose the profiled python script (default: bench.py, one step).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RA_PIPELINE_CHUNKS=1
SCRIPT=${PMC_SCRIPT:-bench.py}
ARGS=${PMC_ARGS:---steps 1 --warmup 1 --no-cpu-baseline}
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python3 $ROOT/$SCRIPT $ARGS > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
cd $ROOT
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pmc_probe")
acc = collections.defaultdict(dict)
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ra_" not in n: continue
        k = n.split("::")[-1].split("(")[0]
        per[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for (k, c), lst in per.items():
        lst.sort()
        acc[k][c] = lst[-1][1]  # last dispatch = the timed step
for k, d in acc.items():
    print(k, {c: f"{v:.4g}" for c, v in sorted(d.items())})
PY
rm -rf $OUT/g*/
