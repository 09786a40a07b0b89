#!/usr/bin/env python3
"""Randomised sweep of SURVEY.md section 8 f2 (GPU box): the band-pass bank vocoder (time-parallel default and the sample-serial pipeline of
reference_order = 1) fed by the carrier synth with all six voices incl. noise, random band counts / limits / follower ratios / channel counts /
envelopes / note lists and block cuts, against the restatement: carrier bit for bit when the sine voice is silent (else 1e-11), audio <= 1e-5 RMS
(<= 1e-8 measured), block cuts to rounding (bit for bit in reference order).

    python scripts/gpu_bank_fuzz.py [n_configs] [seed]
"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api, signals


def notes(rng, n, stream, count):
    rows = []
    for _ in range(count):
        on = int(rng.integers(0, max(1, n - 1500)))
        off = on + int(rng.integers(300, max(301, n // 2)))
        key = int(rng.integers(30, 90))
        rows.append((on, stream, 0, key, int(rng.integers(0, 60)), 0.0, float(rng.uniform(0.2, 1.0))))
        rows.append((off, stream, 1, key, int(rng.integers(0, 60)), 0.0, 0.0))
    rows.sort(key=lambda r: r[0])
    return rows


def run(n_cfg, seed, verbose=True):
    rng = np.random.default_rng(seed)
    failures = []
    for ci in range(n_cfg):
        sr = int(rng.choice([44100, 48000, 8000]))
        ro = int(rng.choice([0, 0, 1]))
        vols = rng.choice([0.0, 0.0, 0.3, 0.6], size=6)
        if not np.any(vols):
            vols[int(rng.integers(0, 6))] = 0.5
        kw = dict(sample_rate=sr, analysis_volume=0.0, stereo_spread=0.0, seed=int(rng.integers(1, 1 << 30)), reference_order=ro, vocoder_kind=1,
                  voice_volume=float(rng.choice([0.0, 0.3])), carrier_volume=float(rng.choice([0.0, 0.2])), vocoder_volume=float(rng.choice([1.0, 3.0])),
                  vocoder_count_bands=int(rng.integers(1, 51)), vocoder_min_freq=float(rng.choice([50.0, 100.0, 300.0])),
                  vocoder_max_freq=float(rng.choice([3000.0, 0.45 * sr])), vocoder_env_follower_cutoff_ratio=float(rng.choice([0.02, 0.05, 0.2, 1.0])),
                  carrier_n_channels=int(rng.choice([1, 3, 8, 16, 40])), vocoder_carrier_noise_volume=float(vols[0]), vocoder_carrier_saw_volume=float(vols[1]),
                  vocoder_carrier_triangle_volume=float(vols[2]), vocoder_carrier_square_volume=float(vols[3]), vocoder_carrier_sine_volume=float(vols[4]),
                  vocoder_carrier_pulse_volume=float(vols[5]), vocoder_carrier_pulse_width=float(rng.choice([0.01, 0.3, 1.0])),
                  env_attack_seconds=float(rng.choice([0.0, 0.01])), env_release_seconds=float(rng.choice([0.0, 0.03])), env_sustain_level=float(rng.choice([1.0, 0.6])),
                  env_decay_seconds=0.02)
        S = int(rng.choice([1, 2, 3]))
        n = int(rng.integers(3000, 40000))
        x = signals.make_streams(S, n, sr=sr, base_seed=int(rng.integers(0, 10000)), kind=str(rng.choice(["partials", "speech"])))
        rows = []
        for s in range(S):
            rows += notes(rng, n, s, int(rng.integers(1, 14)))
        rows.sort(key=lambda r: (r[1], r[0]))
        msg = []
        try:
            with api.Context(api.default_config(**kw), S, 0) as ctx:
                ctx.carrier_events(rows)
                out = ctx.process_host(x)
                cars = [ctx.carrier(n, s) for s in range(S)]
                st = ctx.stats()
                if st.error_flags: msg.append("error flags")
            with api.Context(api.default_config(**kw), S, 0) as ctx:
                ctx.carrier_events(rows)
                cuts = sorted(set([0, n] + [int(c) for c in rng.integers(1, n, 3)]))
                cat = np.concatenate([ctx.process_host(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
            d = float(np.max(np.abs(cat.astype(np.float64) - out.astype(np.float64))))
            if (ro == 1 and d != 0.0) or d > 1e-6: msg.append(f"block cuts change the output by {d:.3g}")
            dropped = 0
            for s in range(S):
                ev = oracle.carrier_events([(r[0], r[2], r[3], r[4], r[5], r[6]) for r in rows if r[1] == s])
                ref, car, _, dr = oracle.run_offline_ex(oracle.default_config(**{k: v for k, v in dict(kw, seed=kw["seed"] + s).items() if k != "reference_order"}), x[s], None, ev, want_carrier=True)
                dropped += dr[1]
                dc = float(np.max(np.abs(cars[s] - car)))
                if (vols[4] == 0.0 and dc != 0.0) or dc > 1e-11: msg.append(f"s{s}: carrier differs by {dc:.3g}")
                e = float(np.sqrt(np.mean((out[s].astype(np.float64) - ref.astype(np.float64)) ** 2)))
                if not (e <= (0.0 if False else 1e-5)): msg.append(f"s{s}: rms {e:.3g}")
                if ro == 1 and not np.array_equal(out[s], ref) and vols[4] == 0.0: msg.append(f"s{s}: reference order is not bit-identical (rms {e:.3g})")
            if st.carrier_dropped_note_on != dropped: msg.append(f"dropped note-ons {st.carrier_dropped_note_on} != {dropped}")
        except api.RaError as e:
            msg.append(f"RaError {e}")
        if msg:
            failures.append((ci, msg, kw))
            if verbose: print(f"config {ci} FAILED: {msg}\n   {kw} S={S} n={n}")
    return failures


if __name__ == "__main__":
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    failures = run(n_cfg, int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    print(f"{n_cfg - len(failures)} of {n_cfg} random bank / carrier-synth configurations agree with the restatement")
    sys.exit(1 if failures else 0)
