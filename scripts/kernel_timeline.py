#!/usr/bin/env python3
"""Tuning aid: prints the launches of one steady-state step from a rocprofv3 --kernel-trace CSV (start / end in ms relative to the step's
first launch, per kernel), so that one can see which queue paces the pipeline.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs
  python scripts/kernel_timeline.py gpurun_out/tl [step_index]
"""
import csv
import glob
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = {"ra_analysis_kernel": "A", "ra_track_kernel": "T", "ra_synth_propagate_kernel": "P", "ra_synth_render_mono_kernel": "R", "ra_synth_render_kernel": "R",
         "ra_roll_history_kernel": "H", "ra_track_prep_kernel": "Q"}
# steps are delimited by the history roll (one per block)
hist = [i for i, r in enumerate(rows) if "roll_history" in r[2]]
per_kind = {}
for s, e, k in rows:
    k = next((v for n, v in short.items() if n in k), None)
    if k:
        per_kind.setdefault(k, []).append((s, e))
# block index of a launch = its ordinal within its kind // launches per block
lpb = {k: len(v) // max(1, len(hist)) for k, v in per_kind.items()}
base = per_kind["A"][which * lpb["A"]][0]
for k in "AQTPR":
    v = per_kind[k][which * lpb[k]:(which + 1) * lpb[k]]
    print(k, " ".join(f"[{(s - base) / 1e6:6.2f} {(e - base) / 1e6:6.2f}]" for s, e in v))
nxt = per_kind["A"][(which + 1) * lpb["A"]][0] if len(per_kind["A"]) > (which + 1) * lpb["A"] else None
if nxt:
    print("next block's first analysis launch starts at", round((nxt - base) / 1e6, 2), "ms")
