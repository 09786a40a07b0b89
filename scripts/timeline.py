"""Tuning aid: prints the kernel timeline of the last bench step from a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    for k in ('ra_analysis', 'ra_track', 'ra_synth_propagate', 'ra_synth_render', 'ra_roll', 'ra_voc'):
        if k in n: return k
    return None
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows if short(r['Kernel_Name']))
rolls = [i for i, e in enumerate(ev) if e[2] == 'ra_roll']
step = ev[rolls[-2] + 1: rolls[-1] + 1]
t0 = min(e[0] for e in step)
busy = {}
for s, e, n in step:
    print(f"{n:20s} {(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e6:6.3f} ms)")
    busy[n] = busy.get(n, 0) + (e - s) / 1e6
print({k: round(v, 2) for k, v in busy.items()}, "total", round((max(e[1] for e in step) - t0) / 1e6, 2))
