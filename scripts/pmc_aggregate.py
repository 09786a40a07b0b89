"""Builds traffic.json (per-kernel PMC sums of the LAST bench step) and kernel_stats.csv (ra_* rows of rocprofv3 --stats) from the
directory scripts/collect_profiles.sh filled."""
import csv, glob, json, os, sys

out = sys.argv[1]
KERNELS = ("ra_analysis_kernel", "ra_track_kernel", "ra_synth_propagate_kernel", "ra_synth_render_mono_kernel", "ra_synth_render_kernel",
           "ra_roll_history_kernel")


def short(name):
    for k in KERNELS:
        if k + "(" in name or name.endswith(k):
            return k
    return None


kern = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    files = glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter file for", ctr)
        continue
    rows = list(csv.DictReader(open(files[0])))
    # one row per dispatch (and counter); the bench ran warmup 1 + steps 1: keep the second half of each kernel's dispatches
    per = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == ctr:
            per.setdefault(k, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, lst in per.items():
        lst.sort()
        half = lst[len(lst) // 2:]
        d = kern.setdefault(k, {})
        d[ctr if ctr == "SQ_INSTS_VALU" else ctr + "_KB"] = sum(v for _, v in half)
        d["launches"] = len(half)
doc = {
    "collected_with": "rocprofv3 --kernel-trace --pmc <one counter per run: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU> -- python3 bench.py --steps 1 "
                      "--warmup 1 --no-cpu-baseline ; sums over the launches of the timed step (scripts/collect_profiles.sh, scripts/pmc_aggregate.py)",
    "units": "FETCH/WRITE in KB as reported by rocprofv3; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 — the x2 is the gfx950 correction of "
             "MI355X_MICROARCH.md §HBM for streaming reads; SQ_INSTS_VALU in wave-instructions",
    "workload": {"streams": 256, "frames_per_stream": 480000, "window": 4096, "stride": 1024, "partials": 64},
    "kernels": kern,
}
json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [r for r in rows[1:] if r and "ra_" in r[0]]
    csv.writer(open(os.path.join(out, "kernel_stats.csv"), "w")).writerows(keep)
print(json.dumps(kern))
