#!/usr/bin/env python3
"""Randomised vocoder sweep (GPU box): random band counts / band limits / modulator windows / strides / mixes and block cuts on short
streams; audio against the restatement (RMS <= 1e-5) and block-cut invariance (bit for bit).

    python scripts/gpu_vocoder_fuzz.py [n_configs] [seed]
"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cpp_audio_amd import api, signals


def run(n_cfg, seed, verbose=True):
    rng = np.random.default_rng(seed)
    failures = []
    for ci in range(n_cfg):
        sr = int(rng.choice([44100, 48000]))
        lo = float(rng.choice([50.0, 100.0, 300.0]))
        kw = dict(sample_rate=sr, analysis_volume=float(rng.choice([0.0, 0.0, 1.0])), stereo_spread=0.0, seed=int(rng.integers(1, 1 << 30)),
                  voice_volume=float(rng.choice([0.0, 0.3])), carrier_volume=float(rng.choice([0.0, 0.2])), vocoder_volume=float(rng.choice([1.0, 0.6])),
                  vocoder_count_bands=int(rng.integers(1, 51)), vocoder_min_freq=lo, vocoder_max_freq=float(rng.choice([3000.0, 8000.0, 20000.0])),
                  vocoder_modulator_window_size_seconds=float(rng.choice([0.005, 0.0213, 0.05, 0.1, 0.17, 0.3])),
                  vocoder_stride_seconds=float(rng.choice([0.0005, 0.0021, 0.005, 0.011, 0.02, 0.04])))
        S = int(rng.choice([1, 2, 5, 5, 2, 1, 1, 2, 5, 48]))  # 48 streams: several modulator frames per workgroup
        n = int(rng.integers(8000, 50000))
        x = signals.make_streams(S, n, sr=sr, base_seed=int(rng.integers(0, 10000)), kind=str(rng.choice(["partials", "speech"])))
        car = np.stack([signals.carrier_stream(n, sr=sr, freq=float(rng.uniform(60.0, 400.0)), kind=str(rng.choice(["saw", "sine"]))) for _ in range(S)])
        msg = []
        try:
            with api.Context(api.default_config(**kw), S, 0) as ctx:
                out = ctx.process_host(x, car)
                ctx.reset()
                cuts = sorted(set([0, n] + [int(c) for c in rng.integers(1, n, 4)]))
                parts = [ctx.process_host(x[:, a:b], car[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
                cat = np.concatenate(parts, axis=1)
                # the vocoder alone is cut-invariant bit for bit; the sine bank's tiles start at the block start, so with it the bar is 1e-6
                if kw["analysis_volume"] == 0.0:
                    if not np.array_equal(cat, out): msg.append("block cuts change the output")
                elif float(np.max(np.abs(cat.astype(np.float64) - out.astype(np.float64)))) > 1e-6: msg.append("block cuts change the output by more than 1e-6")
                if ctx.stats().error_flags: msg.append("error flags")
            for s in range(S):
                ref, _, _ = oracle.run_offline(oracle.default_config(**dict(kw, seed=kw["seed"] + s)), x[s], car[s])
                e = float(np.sqrt(np.mean((out[s].astype(np.float64) - ref.astype(np.float64)) ** 2)))
                if not (e <= 1e-5): msg.append(f"s{s}: rms {e:.3g} (ref rms {float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))):.3g})")
        except api.RaError as e:
            msg.append(f"RaError {e}")
        if msg:
            failures.append((ci, msg, kw))
            if verbose: print(f"config {ci} FAILED: {msg}\n   {kw} S={S} n={n}")
    return failures


if __name__ == "__main__":
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    failures = run(n_cfg, int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
    print(f"{n_cfg - len(failures)} of {n_cfg} random vocoder configurations agree with the restatement")
    sys.exit(1 if failures else 0)
