"""error of each synthesis mode against the oracle and the reference-order kernel (diagnostics)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from cpp_audio_amd import api, signals
n = 120000
x = np.concatenate([signals.make_streams(3, n), signals.make_streams(2, n, kind="speech")])
kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
ref = np.stack([O.run_offline(O.default_config(**dict(kw, seed=1234 + s)), x[s])[0] for s in range(5)]).astype(np.float64)
outs = {}
for mode in (1, 2, 3, 0):
    with api.Context(api.default_config(reference_order=mode, **kw), 5, 0) as ctx:
        outs[mode] = ctx.process_host(x).astype(np.float64)
    d = outs[mode] - ref
    print("mode", mode, "rms vs oracle", np.sqrt(np.mean(d ** 2)), "max", np.max(np.abs(d)), "| max vs mode1", np.max(np.abs(outs[mode] - outs[1])))
