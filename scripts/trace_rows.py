#!/usr/bin/env python3
"""Tuning aid: every ra_* launch of a rocprofv3 --kernel-trace CSV longer than `min_ms`, with start / end / duration in ms from the first one:
   python scripts/trace_rows.py <dir or csv> [min_ms] [name filter]"""
import csv, glob, os, re, sys
src = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flt = sys.argv[3] if len(sys.argv) > 3 else ""
f = src if os.path.isfile(src) else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(ra_[a-z_0-9]+)", r["Kernel_Name"])
    if m and flt in m.group(1):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
rows.sort()
b = rows[0][0]
for s, e, k in rows:
    if (e - s) / 1e6 >= min_ms:
        print(f"{(s - b) / 1e6:9.2f} {(e - b) / 1e6:9.2f} {(e - s) / 1e6:8.2f} {k}")
