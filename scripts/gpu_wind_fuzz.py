#!/usr/bin/env python3
"""Randomised sweep of gen.voice's WIND mode (ra_voice_* against oracle/oracle_wind.hpp + oracle_voice.hpp) on the GPU box: random programs
(gains, filter order 1 .. 258, widths, centre ranges, slow-step counts from one step per noise sample up, short-term noise, every interpolation),
sample rates, notes held and released, random block and chunk cuts."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
bad = 0
worst = 0.0
for ci in range(n_cfg):
    sr = int(rng.choice([8000, 22050, 44100, 48000, 96000]))
    kw = dict(sample_rate=sr, seed=int(rng.integers(1, 1 << 40)), n_channels=int(rng.integers(1, 6)),
              pink_noise_bp_gain=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])), pink_noise_br_gain=float(rng.choice([0.0, 0.0, rng.uniform(0, 1)])),
              sine_gain=float(rng.choice([0.0, 0.1, rng.uniform(0, 1)])), order_filters=int(rng.choice([1, 2, 7, 32, 33, 64, 65, 129, 258, rng.integers(1, 259)])),
              bp_octave_width_min=float(rng.uniform(0, 5)), bp_octave_width_max=float(rng.uniform(0, 5)),
              center_octave_min_long_term=float(rng.uniform(0, 10)), center_octave_max_long_term=float(rng.uniform(0, 10)),
              center_short_term_ratio=float(rng.choice([0.0, rng.uniform(0, 1)])),
              # sample_rate * 2.268^x steps per noise sample: from ~1 step up to seconds
              seconds_slow_iter_long_term=float(rng.uniform(np.log(1.2 / sr) / np.log(2.268), 1.0)),
              seconds_slow_iter_short_term=float(rng.uniform(np.log(1.0 / sr) / np.log(2.268), 0.0)),
              interpolation=int(rng.integers(0, 23)), markov_start_node=int(rng.integers(0, 2)), markov_min_path_length=int(rng.integers(0, 8)), markov_pre_tries=int(rng.integers(0, 4)), markov_additional_tries=int(rng.integers(0, 4)),
              xfade_length_seconds=float(rng.uniform(0.0023, 0.0454)), random_pan=int(rng.integers(0, 2)), pan=float(rng.uniform(-1, 1)),
              loudness_compensation_amount=float(rng.uniform(0, 1)), seed_param=int(rng.choice([0, 1, 48])))
    if kw["pink_noise_bp_gain"] == 0 and kw["pink_noise_br_gain"] == 0 and kw["sine_gain"] == 0:
        kw["pink_noise_bp_gain"] = 1.0
    cfg, _ = api.voice_program(api.VOICE_WIND, 0, **kw)
    S, n = 2, int(rng.integers(3000, 30000))
    rows, nid = [], 0
    for s in range(S):
        t = 0
        for _ in range(int(rng.integers(1, 5))):
            t += int(rng.integers(0, n // 3))
            nid += 1
            rows.append((t, s, 0, nid, float(rng.uniform(50, 3000)), float(rng.uniform(0.2, 1.0))))
            if rng.integers(0, 2):
                rows.append((t + int(rng.integers(1, n)), s, 1, nid, 0.0, 0.0))
    rows.sort(key=lambda r: (r[1], r[0]))
    cuts, left = [], n
    while left > 0:
        m = int(min(left, rng.choice([1, 63, 64, 65, 1000, 4096, n])))
        cuts.append(m); left -= m
    os.environ["RA_WIND_CHUNK"] = str(int(rng.choice([64, 128, 4160, 1 << 20])))
    with api.Voice(cfg, S, 0) as v:
        v.events(rows)
        out = np.concatenate([v.process(m) for m in cuts], axis=1)
        dropped = v.dropped_note_ons()
    ok, ref_dropped, msg = True, 0, ""
    for s in range(S):
        ref, d = oracle.voice_run(oracle.voice_params_from(cfg, cfg.seed + s), [(r[0], r[2], r[3], r[4], r[5]) for r in rows if r[1] == s], n)
        ref_dropped += d
        level = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        err = float(np.sqrt(np.mean((out[s].astype(np.float64) - ref) ** 2)))
        fin = bool(np.all(np.isfinite(out[s])) and np.all(np.isfinite(ref)))
        if level > 0:
            worst = max(worst, err / level)
        if not fin or err > 6e-8 or err > 1e-5 * level + 1e-12:
            ok = False
            msg += f" stream {s}: level {level:.3g} err {err:.3g} finite {fin};"
    if dropped != ref_dropped:
        ok, msg = False, msg + f" dropped {dropped}/{ref_dropped}"
    if not ok:
        bad += 1
        print(f"config {ci} FAILED:{msg} {kw} n {n} cuts {cuts[:6]} chunk {os.environ['RA_WIND_CHUNK']} rows {rows}")
print(f"{n_cfg - bad} of {n_cfg} random WIND configurations agree with the restatement (worst error / level {worst:.3g})")
sys.exit(1 if bad else 0)
