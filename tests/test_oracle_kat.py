"""Pins the CPU oracle against every known-answer test the reference holds for this path
(/root/reference/source/rt.resynth.lib.test.cpp:5-28) and against closed forms / numpy for the
arithmetic the reference outsources to cpp.algorithms (parity unpinned there).
"""
import ctypes as C

import numpy as np
import pytest

DO, DOD, RE, RED, MI, FA, FAD, SOL, SOLD, LA, LAD, SI = range(12)


def _add(o, note, octave, n):
    a, b = C.c_int(), C.c_int()
    o.lib().ora_note_add_halftones(note, octave, n, C.byref(a), C.byref(b))
    return a.value, b.value


def test_kat_note_octave_arithmetic(oracle):
    # rt.resynth.lib.test.cpp:5-13
    assert _add(oracle, FAD, 4, 1) == (SOL, 4)
    assert _add(oracle, FAD, 4, 13) == (SOL, 5)
    assert _add(oracle, FAD, 4, -11) == (SOL, 3)
    d = oracle.lib().ora_note_dist_halftones
    assert d(FAD, 4, SOL, 4) == 1
    assert d(FAD, 4, SOL, 5) == 13
    assert d(FAD, 4, SOL, 3) == -11


def test_kat_major_scale_autotune(oracle):
    # rt.resynth.lib.test.cpp:16-28: F#4 with a C major scale rooted at C4 snaps to G4 (+-1e-4)
    L = oracle.lib()
    pitch = L.ora_midi_get_pitch(FAD, 4)
    expected = L.ora_midi_get_pitch(SOL, 4)
    got = L.ora_scale_closest_pitch(0, L.ora_midi_get_pitch(DO, 4), pitch)
    assert abs(expected - got) < 1e-4


def test_midi_pitch_math(oracle):
    L = oracle.lib()
    assert L.ora_midi_get_pitch(LA, 4) == 69.0
    for f in [27.5, 440.0, 1234.5, 12000.0]:
        p = L.ora_frequency_to_midi_pitch(f)
        assert abs(p - (69 + 12 * np.log2(f / 440.0))) < 1e-12
        assert abs(L.ora_midi_pitch_to_freq(p) - f) < 1e-9 * f
    assert np.isnan(L.ora_frequency_to_midi_pitch(0.0))


def test_explicit_log2_exp10_against_libm(oracle):
    # oracle_math.hpp: the analysis path's log2 / log10 / exp10 as explicit IEEE operation sequences (the product evaluates the same
    # sequences, csrc/ra_math.cuh).  Against libm (numpy): a few ulps, and exact where the result is a power of two.
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-300), np.log(1e12), 200000)), 1.0 + rng.uniform(-1e-3, 1e-3, 20000),
                        2.0 ** np.arange(-60, 60).astype(np.float64)])
    got, ref = oracle.det_math(x, "log2"), np.log2(x)
    ulp = np.spacing(np.maximum(np.abs(ref), 1e-300))
    assert np.max(np.abs(got - ref) / ulp) <= 4.0
    assert np.array_equal(oracle.det_math(2.0 ** np.arange(-60, 60).astype(np.float64), "log2"), np.arange(-60, 60).astype(np.float64))
    got, ref = oracle.det_math(x, "log10"), np.log10(x)
    assert np.max(np.abs(got - ref) / np.spacing(np.maximum(np.abs(ref), 1e-300))) <= 6.0
    y = rng.uniform(-15.0, 5.0, 200000)   # dB / 20 of peak magnitudes
    got, ref = oracle.det_math(y, "exp10"), 10.0 ** y
    assert np.max(np.abs(got - ref) / np.spacing(ref)) <= 3.0
    assert np.array_equal(oracle.det_math(np.array([0.0]), "exp10"), np.array([1.0]))
    pit = rng.uniform(0.0, 140.0, 200000)  # event frequencies: 440 * (2^(1/12))^(pitch - 69) within a few ulps of libm's power
    got, ref = oracle.det_math(pit, "pitch_to_freq"), 440.0 * np.power(2.0 ** (1.0 / 12.0), pit - 69.0)
    assert np.max(np.abs(got - ref) / np.spacing(ref)) <= 16.0
    assert np.mean(got.astype(np.float32) != ref.astype(np.float32)) < 1e-5


@pytest.mark.parametrize("L", [2, 4, 64, 1024, 4096, 8192])
def test_fft_vs_numpy(oracle, L):
    rng = np.random.default_rng(L)
    x = rng.standard_normal(L)
    X = oracle.real_fft(x)
    R = np.fft.rfft(x)
    assert np.max(np.abs(X - R)) < 1e-13 * np.max(np.abs(R)) * np.log2(L + 1)
    assert np.max(np.abs(oracle.real_ifft(X, L) - x)) < 1e-13


def test_window_is_gaussian_4_sigma_normalised(oracle):
    n = 2048
    hw = oracle.half_window(0, n)
    assert abs(hw.sum() - 1.0) < 1e-12          # full window sums to 2
    d = np.arange(n) + 0.5
    g = np.exp(-d * d / (2 * (n / 4.0) ** 2))
    assert np.allclose(hw, g / g.sum(), rtol=1e-13)
    assert hw[-1] / hw[0] < np.exp(-7.9)        # truncated at ~4 sigma


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_peak_picker_on_known_sinusoids(oracle, kind):
    # the test the reference's author wished for (rt.resynth.txt:202-204)
    sr, N = 48000, 4096
    t = np.arange(N) / sr
    comps = [(440.0, 0.25), (1000.3, 0.1), (5021.7, 0.02)]
    x = sum(a * np.sin(2 * np.pi * f * t + 0.3 * i) for i, (f, a) in enumerate(comps))
    sq, L = oracle.sqmag(x, kind)
    f, d = oracle.peaks(sr, sq, L)
    assert np.all(np.diff(f) > 0)               # ascending invariant (algo.cpp:15-21)
    tol_f, tol_a = {0: (0.05, 0.01), 1: (6.0, 0.25), 2: (6.0, 0.6)}[kind]  # dB-parabola is exact for a Gaussian only
    for fr, a in comps:
        i = np.argmin(np.abs(f - fr))
        assert abs(f[i] - fr) < tol_f
        assert abs(10 ** (d[i] / 20) - a) < tol_a * a


def test_sqmag_of_zeros_has_no_peaks(oracle):
    sq, L = oracle.sqmag(np.zeros(1024))
    assert np.all(sq == 0)
    f, d = oracle.peaks(48000, sq, L)
    assert len(f) == 0


def test_pan_law_and_hash(oracle):
    l, r = C.c_double(), C.c_double()
    for pan in [-2, -1, -0.3, 0, 0.5, 1, 3]:
        oracle.lib().ora_stereo(pan, C.byref(l), C.byref(r))
        p = min(max(pan, -1), 1)
        assert abs(l.value - np.cos(np.pi / 4 * (p + 1))) < 1e-15
        assert abs(r.value - np.sin(np.pi / 4 * (p + 1))) < 1e-15
    u = np.array([oracle.lib().ora_hash_uniform_pm1(7, i, 0) for i in range(4000)])
    assert u.min() >= -1 and u.max() < 1 and abs(u.mean()) < 0.05 and abs(u.std() - 3 ** -0.5) < 0.03


def test_sizes_from_seconds(oracle):
    L = oracle.lib()
    # lib.cpp:484-487: the float->int truncation lands on N-2 for an exact N/fs window; (N+0.5)/fs gives N
    assert L.ora_even_window_size_frames(4096.5 / 48000, 48000) == 4096
    assert L.ora_even_window_size_frames(0.1814, 88200) == 15998
    assert L.ora_window_center_stride_frames(1024 / 48000, 48000) == 1024
    assert L.ora_good_stride(0.005, 48000) == 241      # vocoder.cpp:84-92 (odd)
    assert L.ora_modulator_window_size(0.10, 48000) == 4800


def test_autotune_variants(oracle):
    L = oracle.lib()
    c = oracle.default_config(use_autotune=1, autotune_type=0)
    for p, e in [(60.2, 60), (61.1, 62), (65.9, 65), (66.0, 67), (70.9, 71), (71.6, 72), (-3.2, -3.0)]:
        assert abs(L.ora_autotune(C.byref(c), p) - e) < 1e-9, p
    c = oracle.default_config(use_autotune=1, autotune_type=2, autotune_factor=3)
    assert L.ora_autotune(C.byref(c), 7.6) == 9.0 or L.ora_autotune(C.byref(c), 7.6) == 6.0
    assert L.ora_autotune(C.byref(c), 7.4) == 6.0
    c = oracle.default_config(use_autotune=1, autotune_type=1, autotune_chord_frequencies=2)
    # chord 0b10010001 = root, +4, +7 (major triad) on C (72) replicated on octaves
    for p, e in [(60.4, 60), (62.4, 64), (66.4, 67), (70.0, 72)]:
        assert L.ora_autotune(C.byref(c), p) == e, p
    # ties go to the lower neighbour (autotune.cpp:209-216)
    assert L.ora_autotune(C.byref(c), 62.0) == 60
