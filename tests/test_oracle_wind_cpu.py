"""CPU tests of the oracle's WIND parts (oracle/oracle_wind.hpp) against independent restatements in numpy / scipy: the Voss-McCartney pink
noise generator and the sound buffer built from it, and the whole voice with the controls frozen (a band whose centre and width do not move is
a constant-coefficient filter cascade over a known stretch of the buffer)."""
import numpy as np
import pytest
from scipy.signal import lfilter


def white(k):
    """draw k of the pink generator: Box-Muller of the declared counter hash (key "pinknois", salt 9), in Python integers"""
    M = (1 << 64) - 1

    def mix(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    h = mix(0x70696E6B6E6F6973 ^ mix((k * 2 + 9) & M))
    u1 = ((h >> 40) + 1.0) / 16777216.0
    u2 = ((h >> 16) & 0xFFFFFF) / 16777216.0
    return np.float32(np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2))


def voss_mccartney(sr, count):
    """GaussianPinkNoiseAlgo (include/noise.h:82-165) written from its description: `rows` rows of white noise, row 0 redrawn every sample, row i >= 1
    when the counter's lowest set bit is bit i - 1 (the last row takes what is beyond), each row read as the line from its previous to its current value
    over its period 2^i; the sample is the rows' mean"""
    rows = int(sr / 20.0).bit_length()
    f32 = np.float32
    value, prev, at = [f32(0)] * rows, [f32(0)] * rows, [0] * rows
    draws = 0

    def redraw(i, counter):
        nonlocal draws
        at[i], prev[i], value[i] = counter, value[i], white(draws)
        draws += 1

    def row_of(counter):
        i = 1 + ((counter & -counter).bit_length() - 1) if counter else 33
        return min(i, rows - 1)
    counter = 0
    for i in range(rows - 1, 0, -1):
        counter = (1 << (rows - 2)) - (1 << (i - 1))
        redraw(row_of(counter), counter)
    out = np.zeros(count, np.float32)
    for n in range(count):
        counter += 1
        redraw(0, counter)
        redraw(row_of(counter), counter)
        s = f32(0)
        for i in range(rows):
            w = f32(1 + counter - at[i]) / f32(1 << i)
            s = f32(s + f32(f32(w * value[i]) + f32(f32(f32(1) - w) * prev[i])))
        out[n] = s / f32(rows)
    return out


def test_pink_buffer_is_the_smoothed_normalised_voss_mccartney_sequence(oracle):
    sr = 8000   # a short buffer: 160 000 frames, 9 rows (the Python generator above is slow)
    buf, abs_mean = oracle.pink_noise(sr)
    period = int(np.float32(sr) / np.float32(0.05))
    fade = min(10 + period // 10, period)
    assert len(buf) == period
    raw = voss_mccartney(sr, fade + period).astype(np.float64)
    before, own = raw[:fade], raw[fade:].copy()
    part = (np.arange(fade, dtype=np.float32) + np.float32(1)) / np.float32(fade + 1)
    angle = (part * np.float32(np.pi / 2)).astype(np.float64)
    own[-fade:] = np.sin(angle).astype(np.float32) * before + np.cos(angle).astype(np.float32) * own[-fade:]
    f = own.astype(np.float32)
    scale = (np.float32(1) - np.float32(np.finfo(np.float32).eps)) / max(-f.min(), f.max())
    own *= scale
    assert np.array_equal(buf, own)
    assert abs_mean == np.float32(np.abs(own).sum() / len(own)) or abs(abs_mean - np.abs(own).mean()) < 1e-7
    assert np.float32(np.abs(buf).max()) == np.float32(1) - np.float32(np.finfo(np.float32).eps)


def test_pink_buffer_has_a_one_over_f_spectrum_and_a_seamless_loop(oracle):
    buf, abs_mean = oracle.pink_noise(48000)
    assert len(buf) == 960000 and 0.05 < abs_mean < 0.5
    n = 1 << 19
    p = np.abs(np.fft.rfft(buf[:n] * np.hanning(n))) ** 2
    f = np.fft.rfftfreq(n, 1 / 48000)
    level = [10 * np.log10(p[(f >= lo) & (f < 2 * lo)].mean()) for lo in (20, 200, 2000)]
    # -10 dB per decade between 10 Hz (the lowest row's rate) and the rows' common ceiling
    assert 8 < level[0] - level[1] < 13 and 8 < level[1] - level[2] < 13
    # the loop point: the last samples were cross-faded into what the generator produced before the buffer's first sample — the step across the wrap
    # is of the size of the steps inside the buffer
    inside = np.abs(np.diff(buf)).mean()
    assert abs(buf[0] - buf[-1]) < 6 * inside


@pytest.mark.parametrize("order, reject", [(1, False), (4, False), (13, False), (3, True)])
def test_frozen_controls_make_the_voice_a_filter_cascade_over_the_buffer(oracle, order, reject):
    # centre range 10 Hz * 2^5.2 on both ends and a width range of 1.3 octaves on both ends: whatever the slow iterators read, the band stays where it
    # is — low-pass sections at inc * 2^1.3, high-pass sections at inc / 2^1.3 (band-reject: the other way round, in parallel) — and the voice is
    # those one-poles (y += k (x - y), k = pi w / (1 + pi w); high-pass x - y) over the buffer from its middle on, under the envelope's linear attack
    sr = 48000
    p = oracle.OraVoiceParams()
    p.sample_rate, p.n_channels, p.mode, p.seed = sr, 2, 3, 9
    p.length_ms, p.length_exponent, p.xfade_length_seconds = 93.3, 2.0, 0.0499
    p.loudness_level, p.loudness_compensation, p.loudness_ref_freq_index = 30.0, 1.0, 5
    p.interpolation, p.markov_min_path_length, p.freq_scatter, p.length_exponent_scatter = 1, 6, 0.12, 0.5
    p.pink_noise_bp_gain, p.pink_noise_br_gain = (0.0, 1.0) if reject else (1.0, 0.0)
    p.bp_octave_width_min = p.bp_octave_width_max = 1.3
    p.center_octave_min_long_term = p.center_octave_max_long_term = 5.2
    p.seconds_slow_iter_long_term, p.order_filters = -3.0, order
    n = 20000
    out, dropped = oracle.voice_run(p, [(0, 0, 1, 440.0, 0.5)], n)
    assert dropped == 0
    buf, _ = oracle.pink_noise(sr)
    x = buf[(len(buf) // 2 + np.arange(n)) % len(buf)]     # setStartAngle(0): the index before the first step is size / 2 - 1
    inc = np.float64(np.float32(np.exp(np.float64(np.float32(np.log(np.float64(np.float32(2) * (np.float32(10) * np.float32(2) ** np.float32(5.2)) / np.float32(sr))))))))
    wf = 2.0 ** np.float64(np.float32(1.3))
    k = lambda w: np.pi * w / (1 + np.pi * w)
    lp = lambda sig, kk: lfilter([kk], [1, -(1 - kk)], sig)
    hp = lambda sig, kk: sig - lp(sig, kk)
    if not reject:
        y = x
        for _ in range(order):
            y = lp(y, k(inc * wf))
        for _ in range(order):
            y = hp(y, k(inc / wf))
    else:
        a, b = x, x
        for _ in range(order):
            a = lp(a, k(inc / wf))
            b = hp(b, k(inc * wf))
        y = a + b
    A = int(0.5 + 0.0499 * sr)
    j = np.arange(1, n + 1)
    env = np.where(j < A, j / A, 1.0)
    ref = 0.5 * np.cos(np.pi / 4) * y * env                 # velocity 0.5, pan 0 (random_pan = 0)
    got = out[:, 0].astype(np.float64)
    assert np.sqrt(np.mean(ref ** 2)) > 1e-4
    assert np.allclose(out[:, 0], out[:, 1])
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 2e-7 * np.sqrt(np.mean(ref ** 2)) + 1e-10
