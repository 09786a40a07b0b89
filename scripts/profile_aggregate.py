"""Builds profiles/rNN/<workload>/{counters.json, kernel_stats.csv} from what scripts/collect_profiles.sh left in gpurun_out/prof_<tag>/<workload>/:
the rocprofv3 --kernel-trace --stats summary and the --pmc passes of `python3 scripts/profile_workload.py <workload> <steps>`.
Per kernel: launches per step, the average launch duration (all launches), and the counters summed over the LAST step's launches (the last
launches / steps dispatches of the kernel).  The workload line (description, sizes, source hash) comes from the run's own JSON line."""
import csv
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(ra_[a-z0-9_]+)", name)
    return m.group(1) if m else None


for wdir in sorted(glob.glob(os.path.join(src, "*"))):
    if not os.path.isdir(wdir):
        continue
    name = os.path.basename(wdir)
    try:
        line = [l for l in open(os.path.join(wdir, "stats_run.log")) if l.startswith('{"workload"')][-1]
    except Exception:
        print("no run line for", name)
        continue
    run = json.loads(line)
    steps = run["steps"]
    kern = {}
    stats = glob.glob(os.path.join(wdir, "stats", "**", "*kernel_stats.csv"), recursive=True)
    os.makedirs(os.path.join(dst, name), exist_ok=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        keep = [rows[0]] + [r for r in rows[1:] if r and "ra_" in r[0]]
        csv.writer(open(os.path.join(dst, name, "kernel_stats.csv"), "w")).writerows(keep)
        for r in keep[1:]:
            k = short(r[0])
            d = kern.setdefault(k, {})
            d["launches_per_step"] = d.get("launches_per_step", 0) + int(r[1]) / steps
            d["total_ms_per_step"] = d.get("total_ms_per_step", 0.0) + float(r[2]) / 1e6 / steps
        for d in kern.values():
            d["avg_launch_ms"] = d["total_ms_per_step"] / d["launches_per_step"]
    for pdir in sorted(glob.glob(os.path.join(wdir, "pmc_*"))):
        files = glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        per = {}
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            if k:
                per.setdefault((k, r["Counter_Name"]), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for (k, ctr), lst in per.items():
            lst.sort()
            ids = sorted({d for d, _ in lst})
            per_step = max(1, len(ids) // steps)
            last = set(ids[-per_step:])
            d = kern.setdefault(k, {})
            d[ctr + "_KB" if ctr in ("FETCH_SIZE", "WRITE_SIZE") else ctr] = sum(v for i, v in lst if i in last)
            d["pmc_launches_in_last_step"] = per_step
    doc = {
        "collected_with": "scripts/collect_profiles.sh: rocprofv3 --kernel-trace --stats and --kernel-trace --pmc <SQ counters | FETCH_SIZE | WRITE_SIZE> (three passes) -- "
                          "python3 scripts/profile_workload.py %s %d ; counters = sums over the launches of the run's last step" % (name, steps),
        "units": "FETCH/WRITE in KB as rocprofv3 reports them; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: gfx950 counts a wide streaming read at "
                 "half its bytes); SQ_INSTS_VALU in wave-instructions; SQ_WAVES waves launched; SQ_BUSY_CYCLES / SQ_WAVE_CYCLES as reported (WAVE_CYCLES in quad-cycles); "
                 "total_ms_per_step = the kernel's launch durations summed over a step INSIDE the pipeline: launches of different kernels (and of one kernel's chunks) overlap, so the "
                 "figures of a workload add up to more than its step, and an issue fraction formed with them understates a kernel that shares the chip",
        "workload": run,
        "csrc_sha256": run.get("csrc_sha256"),
        "kernels": kern,
    }
    json.dump(doc, open(os.path.join(dst, name, "counters.json"), "w"), indent=1)
    print(name, {k: round(v.get("total_ms_per_step", 0), 3) for k, v in kern.items()})
