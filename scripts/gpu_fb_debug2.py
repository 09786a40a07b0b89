"""Debug aid: feedback with a block split."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from cpp_audio_amd import api, signals
import oracle_lib as oracle
kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234, window_size=1024, stride=256, n_channels=16)
for delay_s, gain, cut in ((0.05, 0.5, 24017), (0.05, 0.5, 24064), (0.05, 0.5, 8000), (0.0, 0.3, 24017)):
    n = 48000
    x = signals.make_streams(1, n, base_seed=31)
    g = api.default_config(analysis_output_feedback_gain=gain, output_delay_seconds=delay_s, **kw)
    o = oracle.default_config(analysis_output_feedback_gain=gain, output_delay_seconds=delay_s, **kw)
    with api.Context(g, 1, 0) as ctx:
        parts = np.concatenate([ctx.process_host(x[:, :cut]), ctx.process_host(x[:, cut:])], axis=1)
    ref, rev, _ = oracle.run_offline(o, x[0])
    d = parts[0].astype(np.float64) - ref
    w = 2000
    print("delay", delay_s, "cut", cut, "rms per 2000:", " ".join(f"{np.sqrt(np.mean(d[i:i + w] ** 2)):.0e}" for i in range(0, n, w)))
