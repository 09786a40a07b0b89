import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from cpp_audio_amd import api, signals as sig
x = np.concatenate([sig.make_streams(3, 120000), sig.make_streams(2, 120000, kind="speech")])
kw = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)
outs = []
for ro in (0, 1):
    with api.Context(api.default_config(reference_order=ro, **kw), 5, 0) as ctx:
        outs.append(ctx.process_host(x).astype(np.float64))
d = outs[0] - outs[1]
print(os.environ.get("RESYNTH_AMD_LIB", "in-tree").split("_")[-1], "rms", np.sqrt(np.mean(d ** 2)), "max", np.max(np.abs(d)))
