"""Seeded synthetic input streams: no dataset ships with the reference.

`partials_stream` : sum of K vibrato partials with gated envelopes + white noise floor.
`speech_like_stream`: gliding harmonic stack, 1/k roll-off.
Both are peak-normalised to 0.5 and returned as float32, 48 kHz mono by default.
"""
import numpy as np


def _gate(n, sr, rng):
    """0/1 gate smoothed by 10 ms ramps: segments of 0.5-2 s on, 0.1-1 s off."""
    g = np.zeros(n, np.float64)
    t = int(rng.uniform(0.0, 0.5) * sr)
    while t < n:
        on = int(rng.uniform(0.5, 2.0) * sr)
        g[t:t + on] = 1.0
        t += on + int(rng.uniform(0.1, 1.0) * sr)
    r = max(1, int(0.01 * sr))
    k = np.ones(r) / r
    return np.convolve(g, k, mode="same")


def partials_stream(seed, n_frames, sr=48000, n_partials=24, noise_db=-60.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n_frames, dtype=np.float64) / sr
    x = np.zeros(n_frames, np.float64)
    for _ in range(n_partials):
        f0 = np.exp(rng.uniform(np.log(80.0), np.log(8000.0)))
        vib_rate = rng.uniform(0.5, 6.0)
        vib_depth = rng.uniform(0.0, 0.3)  # semitones
        amp = 10.0 ** (rng.uniform(-40.0, -6.0) / 20.0)
        ph = rng.uniform(0, 2 * np.pi)
        f = f0 * 2.0 ** (vib_depth / 12.0 * np.sin(2 * np.pi * vib_rate * t + rng.uniform(0, 2 * np.pi)))
        phase = ph + 2 * np.pi * np.cumsum(f) / sr
        x += amp * _gate(n_frames, sr, rng) * np.sin(phase)
    x += 10.0 ** (noise_db / 20.0) * rng.standard_normal(n_frames)
    m = np.max(np.abs(x))
    if m > 0:
        x *= 0.5 / m
    return x.astype(np.float32)


def speech_like_stream(seed, n_frames, sr=48000, n_harmonics=30, noise_db=-60.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n_frames, dtype=np.float64) / sr
    a, b = rng.uniform(90.0, 250.0, 2)
    rate = rng.uniform(0.2, 1.0)
    f0 = a + (b - a) * 0.5 * (1 + np.sin(2 * np.pi * rate * t))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n_frames, np.float64)
    for k in range(1, n_harmonics + 1):
        x += (1.0 / k) * np.sin(k * phase + rng.uniform(0, 2 * np.pi))
    x *= _gate(n_frames, sr, rng)
    x += 10.0 ** (noise_db / 20.0) * rng.standard_normal(n_frames)
    m = np.max(np.abs(x))
    if m > 0:
        x *= 0.5 / m
    return x.astype(np.float32)


def make_streams(n_streams, n_frames, sr=48000, base_seed=1234, kind="partials"):
    f = partials_stream if kind == "partials" else speech_like_stream
    return np.stack([f(base_seed + s, n_frames, sr) for s in range(n_streams)])


def carrier_stream(n_frames, sr=48000, freq=110.0, kind="saw"):
    """Vocoder carrier (config C3): 110 Hz saw (or sine), amplitude 0.5."""
    t = np.arange(n_frames, dtype=np.float64) / sr
    if kind == "sine":
        x = np.sin(2 * np.pi * freq * t)
    else:
        x = 2.0 * ((freq * t) % 1.0) - 1.0
    return (0.5 * x).astype(np.float32)
