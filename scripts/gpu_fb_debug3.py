#!/usr/bin/env python3
"""Debug aid: first divergence between the product and the restatement on one feedback configuration."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api, signals
kw = {'analysis_volume': 1.0, 'stereo_spread': 0.0, 'carrier_volume': 0.0, 'seed': 256744668, 'window_size': 1024, 'stride': 300, 'n_channels': 64,
      'analysis_output_feedback_gain': -0.15, 'output_delay_seconds': 0.0031}
n = 43493
x = signals.make_streams(2, n, base_seed=int(sys.argv[1]) if len(sys.argv) > 1 else 1234)
with api.Context(api.default_config(**kw), 2, 0) as ctx:
    out = ctx.process_host(x)
    for s in range(2):
        ocfg = oracle.default_config(**dict(kw, seed=kw["seed"] + s))
        ref, ev, _ = oracle.run_offline(ocfg, x[s])
        gev = ctx.events(s)
        ev = ev[ev["type"] != 3]
        d = np.abs(out[s].astype(np.float64) - ref.astype(np.float64)).max(axis=1)
        first = int(np.argmax(d > 1e-6)) if (d > 1e-6).any() else -1
        m = min(len(ev), len(gev))
        diff = [i for i in range(m) if any(ev[f][i] != gev[f][i] for f in ("sample", "type", "channel", "noteid", "frequency", "velocity"))]
        i0 = diff[0] if diff else (m if len(ev) != len(gev) else -1)
        print(f"stream {s}: rms {float(np.sqrt(np.mean((out[s] - ref.astype(np.float64)) ** 2))):.3g}; first sample off by > 1e-6: {first}; max |d| before it {float(d[:max(first,1)].max()):.3g}; "
              f"events {len(ev)}/{len(gev)}, first differing event {i0} at sample {int(ev[i0]['sample']) if 0 <= i0 < len(ev) else None}")
        if 0 <= i0 < m:
            for i in range(max(i0 - 2, 0), min(i0 + 4, m)): print("    o", tuple(ev[i]), " g", tuple(gev[i]))
