#!/usr/bin/env python3
"""Debug aid: configs[3]-shaped run on a few streams of the bench signal against the oracle (events, frame pitches, audio)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import oracle_lib as oracle
from cpp_audio_amd import api

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 144000
dev = torch.device("cuda", 0)
voice = bench.synth_inputs_torch(384, n, 48000, 4321, dev)[:S].contiguous()
kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234, window_size=8192, stride=2048, n_channels=128,
          use_autotune=1, autotune_type=0, autotune_scale_mode=0, autotune_root_note=0)
hv = voice.cpu().numpy()
with api.Context(api.default_config(**kw), S, 0) as ctx:
    out = ctx.process_host(hv)
    st = ctx.stats()
    print("error_flags", st.error_flags, "dropped", st.dropped_note_on)
    for s in range(min(S, 3)):
        ocfg = oracle.default_config(**dict(kw, seed=kw["seed"] + s))
        ref, ev, dropped = oracle.run_offline(ocfg, hv[s])
        fire, offs, pitch, vol = oracle.run_frames(ocfg, hv[s])
        gfire, goffs, gpitch, gvol = ctx.frame_pitches(s)
        e = float(np.sqrt(np.mean((out[s].astype(np.float64) - ref.astype(np.float64)) ** 2)))
        same_offs = np.array_equal(offs, goffs)
        first_bad = -1
        if not same_offs:
            m = min(len(offs), len(goffs))
            bad = np.nonzero(offs[:m] != goffs[:m])[0]
            first_bad = int(bad[0]) if len(bad) else m
        dp = float(np.max(np.abs(pitch - gpitch))) if same_offs and len(pitch) else None
        gev = ctx.events(s)
        ev = ev[ev["type"] != 3]
        print(f"stream {s}: rms {e:.3e} frames {len(fire)}/{len(gfire)} offs_equal {same_offs} first_bad_frame {first_bad} max|dpitch| {dp} "
              f"events {len(ev)}/{len(gev)} max list {int(np.max(np.diff(offs))) if len(offs) > 1 else 0} dropped {dropped}")
        if not same_offs and first_bad >= 0:
            f = max(first_bad - 1, 0)
            print("   oracle sizes", np.diff(offs)[f:f + 4], "gpu sizes", np.diff(goffs)[f:f + 4])
        m = min(len(ev), len(gev))
        diff = [i for i in range(m) if any(ev[f][i] != gev[f][i] for f in ("sample", "type", "channel", "noteid", "frequency", "velocity"))]
        if diff or len(ev) != len(gev):
            i0 = diff[0] if diff else m
            print("   first differing event index", i0)
            for i in range(max(i0 - 3, 0), min(i0 + 6, m)):
                print("     o", ev[i], " g", gev[i])
            smp = ev[i0]["sample"] if i0 < len(ev) else -1
            fr = int(np.searchsorted(fire, smp)) if smp >= 0 else -1
            for f in (fr - 1, fr):
                if 0 <= f < len(fire):
                    print(f"   frame {f} fire {fire[f]} pitches", np.round(pitch[offs[f]:offs[f + 1]], 4).tolist())
                    print(f"            vols", np.round(vol[offs[f]:offs[f + 1]], 6).tolist())
        if diff or len(ev) != len(gev):
            nid = ev[i0]["noteid"]
            print("   oracle events of note", nid, [tuple(e) for e in ev if e["noteid"] == nid][:12])
            print("   gpu    events of note", nid, [tuple(e) for e in gev if e["noteid"] == nid][:12])
            lo, hi = fire[max(fr - 2, 0)], fire[fr]
            print("   oracle events in the two frames before:")
            for e in ev:
                if lo <= e["sample"] <= hi: print("      ", tuple(e))
            for f in (fr - 1, fr):
                print(f"   frame {f} oracle hex", [float(x).hex() for x in pitch[offs[f]:offs[f + 1]][:6]])
                print(f"   frame {f} gpu    hex", [float(x).hex() for x in gpitch[goffs[f]:goffs[f + 1]][:6]])
