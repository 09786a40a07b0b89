"""Tuning aid: busy time and gaps per kernel over the last ~2 bench steps of a rocprofv3 --kernel-trace CSV (overlapped blocks)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    for k in ('ra_analysis', 'ra_track', 'ra_synth_propagate', 'ra_synth_render', 'ra_roll'):
        if k in n: return k
    return None
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows if short(r['Kernel_Name']))
rolls = [e for e in ev if e[2] == 'ra_roll']
t_lo, t_hi = rolls[-4][1], rolls[-1][1]        # three steady-state blocks
print("window %.2f ms (3 blocks)" % ((t_hi - t_lo) / 1e6))
for k in ('ra_analysis', 'ra_track', 'ra_synth_propagate', 'ra_synth_render'):
    xs = [(s, e) for s, e, n in ev if n == k and e > t_lo and s < t_hi]
    busy = sum(min(e, t_hi) - max(s, t_lo) for s, e in xs) / 1e6
    gaps = sorted(((xs[i + 1][0] - xs[i][1]) / 1e6 for i in range(len(xs) - 1)), reverse=True)
    print(f"{k:20s} launches {len(xs):3d} busy {busy:7.2f} ms  largest gaps {[round(g, 2) for g in gaps[:6]]}")
