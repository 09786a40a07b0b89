#!/bin/bash
# Tuning aid (GPU box): SQ_INSTS_VALU of the render kernel, summed over one bench run, for each ab_libs variant given as argument.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$ROOT/gpurun_out/abl_$v
  RESYNTH_AMD_LIB=$ROOT/ab_libs/libresynth_$v.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT.log 2>&1
  python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys, collections
acc = collections.Counter(); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ra_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("::")[-1].split("(")[0]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
print(sys.argv[2], {k: (round(v / 1e9, 4), n[k]) for k, v in acc.items() if "render" in k})
PY
  rm -rf $OUT
done
