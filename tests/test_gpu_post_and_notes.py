"""GPU parity tests for SURVEY.md §8 f3: post "limit" (K8) and note deduction (ra_deduce_notes) with driver (A) end to end.
Both restate cpp.algorithms behaviour that is absent from the reference checkout (PARITY UNPINNED): these tests pin
the product to the CPU restatement of the same declared decisions."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=1234)


@pytest.fixture(scope="module")
def api():
    from cpp_audio_amd import api
    assert api.lib().ra_device_count() > 0, "no GPU visible: GPU tests cannot run (there is no CPU fallback)"
    return api


@pytest.fixture(scope="module")
def sig():
    from cpp_audio_amd import signals
    return signals


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.mark.parametrize("kw", [dict(analysis_volume=6.0), dict(analysis_volume=6.0, stereo_spread=1.0),
                                dict(analysis_volume=0.5, voice_volume=3.0), dict(analysis_volume=6.0, reference_order=1)])
def test_post_limit_matches_oracle(api, oracle, sig, kw):
    # loud settings so that the limiter works (peaks well above the 0.95 ceiling before it)
    n, S = 60000, 3
    x = sig.make_streams(S, n, base_seed=808)
    gk = dict(KW, post=1, **kw)
    ok = {k: v for k, v in gk.items() if k != "reference_order"}
    with api.Context(api.default_config(**gk), S, 0) as ctx:
        out = ctx.process_host(x)
        assert ctx.stats().error_flags == 0
        ctx.reset()
        parts = np.concatenate([ctx.process_host(x[:, :20011]), ctx.process_host(x[:, 20011:])], axis=1)   # the limiter's state crosses blocks
    assert rms(parts, out) < 1e-7
    worked = False
    for s in range(S):
        ref, _, _ = oracle.run_offline(oracle.default_config(**dict(ok, seed=1234 + s)), x[s])
        raw, _, _ = oracle.run_offline(oracle.default_config(**dict(ok, seed=1234 + s, post=0)), x[s])
        worked |= np.max(np.abs(raw)) > 1.2 and rms(raw, ref) > 1e-2
        assert np.max(np.abs(ref)) <= 1.0
        assert rms(out[s], ref) <= 1e-5, f"stream {s}: {rms(out[s], ref):.3g}"
    assert worked, "the limiter never had to work: the test would be vacuous"


def test_post_limit_publishes_the_limited_signal_to_the_feedback(api, oracle, sig):
    # init_post (lib.cpp:1245-1275): the limiter runs before the mono publication, so the analysis is fed the limited output
    n = 48000
    x = sig.make_streams(2, n, base_seed=5)
    gk = dict(KW, analysis_volume=5.0, post=1, window_size=1024, stride=256, n_channels=16, analysis_output_feedback_gain=0.1, output_delay_seconds=0.02,
              reference_order=1)
    with api.Context(api.default_config(**gk), 2, 0) as ctx:
        out = ctx.process_host(x)
    ok = {k: v for k, v in gk.items() if k != "reference_order"}
    for s in range(2):
        ref, _, _ = oracle.run_offline(oracle.default_config(**dict(ok, seed=1234 + s)), x[s])
        assert rms(out[s], ref) < 1e-6


def three_tone_signal(sr, seconds=2.0):
    t = np.arange(int(sr * seconds)) / sr
    x = np.zeros_like(t)
    for f, a, t0, t1 in ((440.0, 0.3, 0.1, 1.2), (660.0, 0.2, 0.5, 1.8), (1500.0, 0.1, 0.0, 0.7), (1580.0, 0.05, 1.0, 1.9)):
        g = np.clip(np.minimum(t - t0, t1 - t) / 0.02, 0.0, 1.0)
        x += a * g * np.sin(2 * np.pi * f * t)
    rng = np.random.default_rng(1)
    x += 1e-4 * rng.standard_normal(len(x))
    return x.astype(np.float32)


def test_deduce_notes_matches_oracle(api, oracle):
    sr = 44100
    x = three_tone_signal(sr)
    g = api.deduce_notes(x, sr, 400, 400)
    st, en, fr, am = oracle.deduce_notes(x, sr, 400, 400)
    assert len(g) == len(st) and len(g) > 4
    assert np.array_equal(g["start_frame"], st) and np.array_equal(g["end_frame"], en)
    assert np.max(np.abs(g["frequency"] / fr - 1.0)) < 1e-12 and np.max(np.abs(g["amplitude_db"] - am)) < 1e-10
    # the four tones are found, at the right place: the loudest long note near each frequency
    for f, t0, t1 in ((440.0, 0.1, 1.2), (660.0, 0.5, 1.8), (1500.0, 0.0, 0.7), (1580.0, 1.0, 1.9)):
        near = g[(np.abs(g["frequency"] - f) < 8.0) & (g["end_frame"] - g["start_frame"] > 20)]
        assert len(near) >= 1, f
        best = near[np.argmax(near["amplitude_db"])]
        assert abs(best["start_frame"] * 400 / sr - t0) < 0.05 and abs(best["end_frame"] * 400 / sr - t1) < 0.05, (f, best)


def test_deduce_notes_batch_equals_one_call_per_signal_and_the_oracle(api, oracle):
    # ra_deduce_notes_batch: many signals through ONE context (one analysis launch sequence per block over all of them, a linker thread per
    # signal); a long signal is cut into blocks of 2^18 frames, which must not show.  Same notes as one call per signal, and as the oracle.
    sr = 44100
    base = three_tone_signal(sr, 7.0)                     # 308 700 frames: two blocks
    rng = np.random.default_rng(3)
    sigs = np.stack([np.roll(base, 997 * k) * (0.5 + 0.1 * k) + 1e-4 * rng.standard_normal(len(base)).astype(np.float32) for k in range(5)]).astype(np.float32)
    got = api.deduce_notes_batch(sigs, sr, 400, 400)
    assert len(got) == 5
    for k in range(5):
        one = api.deduce_notes(sigs[k], sr, 400, 400)
        assert len(got[k]) == len(one) > 4 and np.array_equal(got[k], one)
    st, en, fr, am = oracle.deduce_notes(sigs[2], sr, 400, 400)
    g = got[2]
    assert len(g) == len(st)
    assert np.array_equal(g["start_frame"], st) and np.array_equal(g["end_frame"], en)
    assert np.max(np.abs(g["frequency"] / fr - 1.0)) < 1e-12 and np.max(np.abs(g["amplitude_db"] - am)) < 1e-10
    with pytest.raises(api.RaError):
        api.deduce_notes_batch(np.zeros((0, 10), np.float32), sr, 400, 400)


def test_driver_a_end_to_end_wav_to_notes_to_wav(api, oracle, tmp_path, run_program):
    # main.resynth (source/main.resynth.cpp:5-81): WAV -> deduceNotesSlow -> 60 dB filter -> resynth -> WAV, all on the HIP backend
    from scipy.io import wavfile
    sr = 44100
    x = three_tone_signal(sr, 1.5)
    wavfile.write(tmp_path / "in.wav", sr, x)
    lib_dir = os.path.join(ROOT, "cpp_audio_amd")
    exe = tmp_path / "resynth_wav"
    rc, so, se = run_program(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "resynth_wav.cpp"), "-o", str(exe),
                              "-L", lib_dir, "-lresynth_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert rc == 0, so + se
    rc, so, se = run_program([str(exe), str(tmp_path / "in.wav"), str(tmp_path / "out.wav")])
    assert rc == 0, so + se
    rendered = np.array([[float(v) for v in l.split()] for l in so.strip().splitlines()])
    # the same flow through the CPU restatement
    st, en, fr, am = oracle.deduce_notes(x, sr, 400, 400)
    keep = (am >= am.max() - 60.0) & (fr != 0)
    assert len(rendered) == int(keep.sum()) and len(rendered) >= 4
    assert np.array_equal(rendered[:, 0], st[keep]) and np.array_equal(rendered[:, 1], en[keep])
    ref = oracle.resynth_notes(st[keep], en[keep], fr[keep], am[keep], 400, sr)
    # the reference writes a mono IEEE float64 WAV (resynth.hpp:29-35)
    from cpp_audio_amd import api as _api  # noqa: F401
    import struct
    raw = open(tmp_path / "out.wav", "rb").read()
    i = raw.index(b"data")
    (nbytes,) = struct.unpack("<I", raw[i + 4:i + 8])
    got = np.frombuffer(raw[i + 8:i + 8 + nbytes], "<f8")
    assert len(got) == len(ref)
    assert np.sqrt(np.mean(ref ** 2)) > 1e-3
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-9
