"""First-light check on the GPU box: kernel-level parity against the oracle, with diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from cpp_audio_amd import api, signals

rng = np.random.default_rng(0)
# 1. sqmag
for N in (800, 1024, 4096, 8192):
    w = rng.standard_normal((3, N)).astype(np.float32)
    g, L = api.sqmag(w)
    for i in range(3):
        r, L2 = O.sqmag(w[i].astype(np.float64))
        err = np.max(np.abs(g[i] - r)) / np.max(r)
        print("sqmag N", N, "L", L, "rel err", err, "bit-exact", np.array_equal(g[i], r))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
S = 4
x = signals.make_streams(S, n)
kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
ctx = api.Context(api.default_config(**kw), S, 0)
t0 = time.time()
out = ctx.process_host(x)
print("gpu time", time.time() - t0)
st = ctx.stats()
print("stats frames", st.frames_analysed, "dropped", st.dropped_note_on, "flags", st.error_flags)
for s in range(S):
    cfg = O.default_config(**dict(kw, seed=1234 + s))
    ref, ev, dropped = O.run_offline(cfg, x[s])
    fire, offs, pitch, vol = O.run_frames(cfg, x[s])
    gfire, goffs, gpitch, gvol = ctx.frame_pitches(s)
    ok_f = np.array_equal(fire, gfire) and np.array_equal(offs, goffs)
    print("stream", s, "frames equal", ok_f, len(fire), len(gfire))
    if ok_f:
        print("   pitch max abs diff", np.max(np.abs(pitch - gpitch)) if len(pitch) else 0, "vol rel", np.max(np.abs(vol - gvol) / vol) if len(vol) else 0)
    gev = ctx.events(s)
    ev = ev[ev["type"] != 3]
    same = len(ev) == len(gev) and all(np.array_equal(ev[k], gev[k]) for k in ("sample", "type", "channel", "noteid", "frequency", "velocity"))
    print("   events", len(ev), len(gev), "identical", same, "dropped", dropped)
    if not same:
        m = min(len(ev), len(gev))
        for i in range(m):
            if ev[i] != gev[i]:
                print("   first diff at", i, ev[i], gev[i]); break
    d = out[s].astype(np.float64) - ref
    print("   rms err", np.sqrt(np.mean(d ** 2)), "max", np.max(np.abs(d)), "ref rms", np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
