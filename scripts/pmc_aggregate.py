"""Builds traffic.json (per-kernel PMC sums of the LAST bench step) and kernel_stats.csv (ra_* rows of rocprofv3 --stats) from the
directory scripts/collect_profiles.sh filled.  The workload block and the kernel-source hash come from the bench line collected
beside the counters (bench.json): bench.py quotes the counters only for exactly that shape and build."""
import csv, glob, json, os, sys

out = sys.argv[1]
STAGE = {"ra_analysis_kernel": "analysis", "ra_track_prep_kernel": "prep",  # (the pre-pass is launched behind every analysis launch; a stage of its own since r04)
          "ra_track_kernel": "track", "ra_synth_propagate_kernel": "propagate", "ra_synth_render_mono_kernel": "render",
         "ra_synth_render_kernel": "render", "ra_roll_history_kernel": "history"}
KERNELS = tuple(STAGE)


def short(name):
    for k in KERNELS:
        if k + "(" in name or k + "<" in name or name.endswith(k):  # templated kernels print as name<256>(...)
            return k
    return None


# the full result of the stats run (bench.py --details): launches per step of every stage, the workload and the source hash
bench = json.load(open(os.path.join(out, "stats_details.json")))
LAUNCHES = {k: v["launches_per_step"] for k, v in bench["kernels"].items()}
cfgw = bench["config"]
import re
m = re.search(r"window (\d+) .*stride (\d+) .* (\d+) partials", cfgw["workload"])
workload = {"streams": cfgw["streams_per_gpu"], "frames_per_stream": cfgw["frames_per_stream_per_step"], "window": int(m.group(1)),
            "stride": int(m.group(2)), "partials": int(m.group(3))}
kern = {}
counters = sorted(os.path.basename(d)[4:] for d in glob.glob(os.path.join(out, "pmc_*")) if os.path.isdir(d))
for ctr in counters:
    files = glob.glob(os.path.join(out, "pmc_" + ctr, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter file for", ctr)
        continue
    rows = list(csv.DictReader(open(files[0])))
    # one row per dispatch (and counter).  The bench ran its set-up block, 1 warm-up step and 1 timed step (--no-alone: no serialised extra
    # steps): the timed step's dispatches of a kernel are its last launches_per_step
    per = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == ctr:
            per.setdefault(k, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, lst in per.items():
        lst.sort()
        nl = int(round(LAUNCHES.get(STAGE[k], 0))) or 1
        last = lst[-nl:]
        d = kern.setdefault(k, {})
        d[ctr + "_KB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else ctr] = sum(v for _, v in last)
        d["launches"] = len(last)
doc = {
    "collected_with": "rocprofv3 --kernel-trace --pmc <one counter per run> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "
                      "--no-other-configs --no-alone ; sums over the launches of the timed step (scripts/collect_profiles.sh, scripts/pmc_aggregate.py)",
    "units": "FETCH/WRITE in KB as reported by rocprofv3; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 — the x2 is the gfx950 correction of "
             "MI355X_MICROARCH.md §HBM for streaming reads; SQ_INSTS_VALU in wave-instructions; SQ_WAVES waves launched; SQ_BUSY_CYCLES / "
             "SQ_WAVE_CYCLES summed over SEs as rocprofv3 reports them; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE in cycles",
    "workload": workload,
    "csrc_sha256": bench.get("csrc_sha256"),
    "kernels": kern,
}
# the clock the chip held inside the two throughput kernels: the diagnostic build's stamps (bench.py --clock-stamps), same sources
try:
    clk = json.load(open(os.path.join(out, "clock_details.json")))["valu"]["effective_clock_ghz_measured"]
    if clk.get("csrc_sha256") == doc["csrc_sha256"]:
        doc["effective_clock_ghz"] = clk
        json.dump(clk, open(os.path.join(out, "clock.json"), "w"), indent=1)
    else:
        print("clock stamps are from another build: not merged")
except Exception as e:
    print("no clock stamps:", e)
json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    keep = [rows[0]] + [r for r in rows[1:] if r and "ra_" in r[0]]
    csv.writer(open(os.path.join(out, "kernel_stats.csv"), "w")).writerows(keep)
print(json.dumps(kern))
