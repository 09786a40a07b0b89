"""GPU parity test for SURVEY.md §8 f4 (first slice): gen.voice in SWEEP mode (K9) against the CPU restatement (oracle_voice.hpp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from cpp_audio_amd import api
    assert api.lib().ra_device_count() > 0, "no GPU visible: GPU tests cannot run (there is no CPU fallback)"
    return api


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def ora_params(oracle, cfg, seed):
    return oracle.OraVoiceParams(cfg.sample_rate, cfg.n_channels, cfg.length_ms, cfg.length_exponent, cfg.xfade_length_seconds, cfg.low_freq, cfg.high_freq,
                                 cfg.sine_gain, cfg.loudness_level, cfg.loudness_compensation_amount, cfg.loudness_ref_freq_index, cfg.pan, cfg.random_pan, seed)


@pytest.mark.parametrize("kw", [dict(), dict(length_ms=40.0, length_exponent=3.0, low_freq=8000.0, high_freq=60.0, pan=-0.6),
                                dict(n_channels=3, random_pan=1, loudness_level=70.0, loudness_compensation_amount=0.5, loudness_ref_freq_index=11, sine_gain=0.5),
                                dict(sample_rate=44100, length_ms=25.0, length_exponent=1.0, xfade_length_seconds=0.0023, low_freq=300.0, high_freq=300.0)])
def test_sweep_voice_matches_oracle(api, oracle, kw):
    cfg = api.voice_default_config(seed=11, **kw)
    S, n = 3, 80000
    rng = np.random.default_rng(5)
    rows = []
    nid = 0
    for s in range(S):
        t = 0
        for _ in range(9):
            t += int(rng.integers(200, 9000))
            nid += 1
            rows.append((t, s, 0, nid, float(rng.uniform(100, 2000)), float(rng.uniform(0.2, 1.0))))
            # some notes are released before their sweep ends (the element falls silent at once and its channel is lost), some after
            rows.append((t + int(rng.integers(300, 60000)), s, 1, nid, 0.0, 0.0))
    rows.sort(key=lambda r: (r[1], r[0]))
    with api.Voice(cfg, S, 0) as v:
        v.events(rows)
        parts = [v.process(m) for m in (1, 7000, 33, n - 7034)]   # block cuts do not matter
        dropped = v.dropped_note_ons()
    out = np.concatenate(parts, axis=1)
    ref_dropped = 0
    for s in range(S):
        ref, d = oracle.voice_run(ora_params(oracle, cfg, 11 + s), [(r[0], r[2], r[3], r[4], r[5]) for r in rows if r[1] == s], n)
        ref_dropped += d
        assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3, "the oracle's voice is silent"
        assert rms(out[s], ref) <= 1e-5, f"stream {s}: {rms(out[s], ref):.3g}"
        assert np.max(np.abs(out[s].astype(np.float64) - ref)) < 1e-5
    assert dropped == ref_dropped
    if cfg.n_channels == 3:
        assert dropped > 0, "three channels and lost channels: some note-ons must be dropped"


def test_sweep_voice_is_a_sweep(api):
    # a 100 Hz -> 10 kHz sweep of 73 * 2^2 ms: the zero crossings get denser, the stereo gains follow the pan law, the level the envelope
    cfg = api.voice_default_config(length_exponent=2.0, high_freq=10000.0, pan=0.5)
    with api.Voice(cfg, 1, 0) as v:
        v.events([(0, 0, 0, 1, 440.0, 1.0)])
        out = v.process(20000)[0]
    dur = int(0.5 + 48 * 73 * 4)
    assert np.all(out[dur + 40:] == 0) and np.any(out[dur - 700:dur] != 0)
    z = np.nonzero(np.diff(np.sign(out[:dur, 0])))[0]
    assert np.mean(np.diff(z[:20])) > 8 * np.mean(np.diff(z[-20:]))
    g = np.cos(np.pi / 4 * 1.5), np.sin(np.pi / 4 * 1.5)
    m = np.abs(out[:, 0]) > 1e-3
    assert np.allclose(out[m, 1] / out[m, 0], g[1] / g[0], rtol=1e-4)


def test_voice_output_resident_in_hbm_equals_host_output(api):
    import torch
    rows = [(100 * k + 37 * s, s, 0, 10 * s + k, 200.0 + 50 * k, 0.8) for s in range(3) for k in range(5)]
    rows.sort(key=lambda r: (r[1], r[0]))
    cfg = api.voice_default_config(seed=5, random_pan=1)
    with api.Voice(cfg, 3, 0) as v:
        v.events(rows)
        host = v.process(30000)
    with api.Voice(cfg, 3, 0) as v:
        v.events(rows)
        out = torch.empty(3, 30000, 2, dtype=torch.float32, device="cuda:0")
        v.process_device(30000, out.data_ptr())
    assert np.any(host != 0) and np.array_equal(out.cpu().numpy(), host)


def test_one_note_per_lane_equals_the_sample_serial_kernel(api, monkeypatch):
    # the default renders every note of a block in a lane of its own and walks the channel bookkeeping on the host; RA_VOICE_SERIAL=1 (read
    # when the object is created) keeps the one-wave-per-synth kernel.  Same samples (the sum over a frame's notes associates differently),
    # same dropped note-ons — with three channels, early releases (lost channels) and long notes that span blocks.
    rng = np.random.default_rng(17)
    S, n = 4, 90000
    rows, nid = [], 0
    for s in range(S):
        t = 0
        for _ in range(30):
            t += int(rng.integers(100, 4000))
            nid += 1
            rows.append((t, s, 0, nid, 440.0, float(rng.uniform(0.2, 1.0))))
            rows.append((t + int(rng.integers(1, 30000)), s, 1, nid, 0.0, 0.0))
    rows.sort(key=lambda r: (r[1], r[0]))
    cfg = api.voice_default_config(seed=3, random_pan=1, n_channels=3, length_exponent=1.5)
    res = []
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("RA_VOICE_SERIAL", "1")
        else:
            monkeypatch.delenv("RA_VOICE_SERIAL", raising=False)
        with api.Voice(cfg, S, 0) as v:
            v.events(rows)
            out = np.concatenate([v.process(m) for m in (5, 20000, 1, 30000, n - 50006)], axis=1)
            res.append((out.astype(np.float64), v.dropped_note_ons()))
    monkeypatch.delenv("RA_VOICE_SERIAL", raising=False)
    assert np.sqrt(np.mean(res[1][0] ** 2)) > 1e-3 and res[1][1] > 0
    assert res[0][1] == res[1][1]
    assert np.max(np.abs(res[0][0] - res[1][0])) < 2e-7


def test_voice_argument_errors(api):
    for kw in (dict(n_channels=0), dict(low_freq=0.0), dict(length_ms=0.0), dict(loudness_ref_freq_index=16), dict(length_ms=10.0, length_exponent=0.0, xfade_length_seconds=0.04)):
        with pytest.raises(api.RaError) as e:
            api.Voice(api.voice_default_config(**kw), 1, 0)
        assert e.value.code == -1
    with api.Voice(api.voice_default_config(), 1, 0) as v:
        for bad in ([(0, 1, 0, 1, 440.0, 1.0)], [(0, 0, 2, 1, 440.0, 1.0)], [(0, 0, 0, 1, 0.0, 1.0)], [(0, 0, 0, 1, 440.0, 0.0)]):
            with pytest.raises(api.RaError):
                v.events(bad)
