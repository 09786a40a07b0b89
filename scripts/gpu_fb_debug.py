"""Tuning/debug aid: GPU vs oracle with output feedback."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from cpp_audio_amd import api, signals
import oracle_lib as oracle
kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234, window_size=1024, stride=256, n_channels=16)
for delay_s, gain in ((0.0, 0.3), (0.05, 0.5)):
    n = 20000
    x = signals.make_streams(1, n, base_seed=31)
    g = api.default_config(analysis_output_feedback_gain=gain, output_delay_seconds=delay_s, **kw)
    o = oracle.default_config(analysis_output_feedback_gain=gain, output_delay_seconds=delay_s, **kw)
    with api.Context(g, 1, 0) as ctx:
        out = ctx.process_host(x)
        ev = ctx.events(0)
    ref, rev, _ = oracle.run_offline(o, x[0])
    rev = rev[rev["type"] != 3]
    d = out[0].astype(np.float64) - ref
    print("delay", delay_s, "gain", gain, "rms", np.sqrt(np.mean(d ** 2)), "n events", len(ev), len(rev))
    bad = np.nonzero(np.abs(d).max(axis=1) > 1e-6)[0]
    print(" first sample with |diff| > 1e-6:", bad[:1], " of", n)
    m = min(len(ev), len(rev))
    for i in range(m):
        if ev["sample"][i] != rev["sample"][i] or ev["type"][i] != rev["type"][i] or abs(ev["frequency"][i] - rev["frequency"][i]) > 1e-6 * max(1, rev["frequency"][i]) or abs(ev["velocity"][i] - rev["velocity"][i]) > 1e-6:
            print(" first differing event", i, ev[i], rev[i])
            break
    w = 1000
    print(" rms per 1000 samples:", " ".join(f"{np.sqrt(np.mean(d[i:i + w] ** 2)):.1e}" for i in range(0, n, w)))
    print(" ref rms per 1000    :", " ".join(f"{np.sqrt(np.mean(ref[i:i + w] ** 2)):.1e}" for i in range(0, n, w)))
