#!/bin/bash
# Tuning aid (runs ON THE GPU BOX): SQ counters and launch durations of every ra_* kernel of one bench workload, serialised by the profiler.
#   gpurun -- 'scripts/pmc_kernels.sh voc_fft [steps]'
W=${1:-configs1}; N=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmck_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -- python3 $ROOT/scripts/profile_workload.py $W $N > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $ROOT
python3 - $OUT $N <<'PY'
import csv, glob, collections, re, sys
out, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(f)):
    m = re.search(r"(ra_[a-z_0-9]+)", r["Kernel_Name"])
    if not m: continue
    k = m.group(1)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(acc.items(), key=lambda kv: -dur[kv[0]]):
    ms = dur[k] / steps
    iv = v.get("SQ_INSTS_VALU", 0) / steps
    print(f"{k:34s} {ms:8.2f} ms/step  VALU {iv / 1e6:9.1f} M  issue {iv / (ms * 1e-3) / 614.4e9 if ms else 0:5.2f}  " + " ".join(f"{a[3:]}={b / steps / 1e6:.0f}M" for a, b in sorted(v.items()) if a != "SQ_INSTS_VALU"))
PY
