"""Builds traffic.json (per-kernel PMC sums of the LAST bench step) and kernel_stats.csv (ra_* rows of rocprofv3 --stats) from the
directory scripts/collect_profiles.sh filled."""
import csv, glob, json, os, sys

out = sys.argv[1]
KERNELS = ("ra_analysis_kernel", "ra_track_kernel", "ra_synth_propagate_kernel", "ra_synth_render_mono_kernel", "ra_synth_render_kernel",
           "ra_roll_history_kernel")


def short(name):
    for k in KERNELS:
        if k + "(" in name or k + "<" in name or name.endswith(k):  # templated kernels print as name<256>(...)
            return k
    return None


kern = {}
STAGE = {"ra_analysis_kernel": "analysis", "ra_track_kernel": "track", "ra_synth_propagate_kernel": "propagate", "ra_synth_render_mono_kernel": "render",
         "ra_synth_render_kernel": "render", "ra_roll_history_kernel": "history"}
try:
    LAUNCHES = json.load(open(os.path.join(out, "bench.json")))["kernel_launches_per_step"]
except Exception:
    LAUNCHES = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    files = glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter file for", ctr)
        continue
    rows = list(csv.DictReader(open(files[0])))
    # one row per dispatch (and counter).  The bench ran its set-up block, 1 warm-up step and 1 timed step: keep the dispatches of the
    # timed step = the last kernel_launches_per_step[stage] of each kernel (from the bench line collected beside this)
    per = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == ctr:
            per.setdefault(k, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, lst in per.items():
        lst.sort()
        nl = int(round(LAUNCHES.get(STAGE[k], len(lst) // 2))) or 1
        last = lst[-nl:]
        d = kern.setdefault(k, {})
        d[ctr if ctr == "SQ_INSTS_VALU" else ctr + "_KB"] = sum(v for _, v in last)
        d["launches"] = len(last)
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
