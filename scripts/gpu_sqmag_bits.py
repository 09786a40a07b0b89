#!/usr/bin/env python3
"""Debug aid (GPU box): are the |X|^2 spectra of the product and the oracle the same bits?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api
rng = np.random.default_rng(1)
for N, kind in ((16, 0), (1024, 0), (4096, 0), (4096, 2), (4800, 0), (8192, 0)):
    w = (0.3 * rng.standard_normal((4, N))).astype(np.float32)
    g, L = api.sqmag(w, kind)
    bad = 0; worst = 0.0
    for i in range(4):
        r, _ = oracle.sqmag(w[i].astype(np.float64), kind)
        ne = g[i] != r
        bad += int(ne.sum())
        if ne.any(): print('   window', i, 'bins', np.nonzero(ne)[0].tolist(), 'gpu', [float(v).hex() for v in g[i][ne]], 'oracle', [float(v).hex() for v in r[ne]])
        if ne.any(): worst = max(worst, float(np.max(np.abs(g[i][ne] - r[ne]) / np.maximum(r[ne], 1e-300))))
    print(f"N={N} kind={kind}: {bad} of {4 * len(r)} bins differ, worst relative {worst:.2e}")
