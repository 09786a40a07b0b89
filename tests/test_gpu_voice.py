"""GPU parity tests for SURVEY.md §8 f4: gen.voice — SWEEP, BIRDS, ROBOTS (K9) and WIND (K10) against the CPU restatement (oracle_voice.hpp, oracle_wind.hpp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# measured on MI355X: the worst stream of these 24 cases is 5.6e-9 RMS; the bar is ten times that
RMS_BAR = 6e-8
PEAK_BAR = 1e-6


@pytest.fixture(scope="module")
def api():
    from cpp_audio_amd import api
    assert api.lib().ra_device_count() > 0, "no GPU visible: GPU tests cannot run (there is no CPU fallback)"
    return api


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def note_rows(rng, S, count, gap, hold, freq=(100, 2000)):
    rows, nid = [], 0
    for s in range(S):
        t = 0
        for _ in range(count):
            t += int(rng.integers(*gap))
            nid += 1
            rows.append((t, s, 0, nid, float(rng.uniform(*freq)), float(rng.uniform(0.2, 1.0))))
            # some notes are released before they have played out (the element falls silent at once and its channel is lost), some after
            rows.append((t + int(rng.integers(*hold)), s, 1, nid, 0.0, 0.0))
    rows.sort(key=lambda r: (r[1], r[0]))
    return rows


def check_against_oracle(api, oracle, cfg, rows, S, n, cuts, seed, expect_sound=True, min_rms=2e-4, rel_bar=None):
    with api.Voice(cfg, S, 0) as v:
        v.events(rows)
        parts = [v.process(m) for m in cuts]
        dropped = v.dropped_note_ons()
    out = np.concatenate(parts, axis=1)
    assert out.shape[1] == n
    ref_dropped, worst = 0, 0.0
    for s in range(S):
        ref, d = oracle.voice_run(oracle.voice_params_from(cfg, seed + s), [(r[0], r[2], r[3], r[4], r[5]) for r in rows if r[1] == s], n)
        ref_dropped += d
        if expect_sound:
            assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > min_rms, "the oracle's voice is silent"
        e = rms(out[s], ref)
        worst = max(worst, e)
        assert e <= RMS_BAR, f"stream {s}: rms {e:.3g}"
        if rel_bar is not None:   # quiet voices: the error relative to the signal as well
            level = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
            print(f"stream {s}: level {level:.3g}, rms error {e:.3g} = {e / level:.3g} of it")
            assert e <= rel_bar * level, f"stream {s}: rms {e:.3g} against a level of {level:.3g}"
        assert np.max(np.abs(out[s].astype(np.float64) - ref)) < PEAK_BAR
    assert dropped == ref_dropped
    print(f"worst stream rms vs oracle: {worst:.3g}")
    return out, dropped, worst


@pytest.mark.parametrize("kw", [dict(), dict(length_ms=40.0, length_exponent=3.0, low_freq=8000.0, high_freq=60.0, pan=-0.6),
                                dict(n_channels=3, random_pan=1, loudness_level=70.0, loudness_compensation_amount=0.5, loudness_ref_freq_index=11, sine_gain=0.5),
                                dict(sample_rate=44100, length_ms=25.0, length_exponent=1.0, xfade_length_seconds=0.0023, low_freq=300.0, high_freq=300.0),
                                dict(interpolation=17, length_exponent=2.0, low_freq=200.0, high_freq=6000.0),   # EASE_IN_EXPO
                                dict(interpolation=1, length_exponent=2.5, low_freq=5000.0, high_freq=50.0)])     # PROPORTIONAL_VALUE_DERIVATIVE
def test_sweep_voice_matches_oracle(api, oracle, kw):
    cfg = api.voice_default_config(seed=11, **kw)
    S, n = 3, 80000
    rows = note_rows(np.random.default_rng(5), S, 9, (200, 9000), (300, 60000))
    _, dropped, _ = check_against_oracle(api, oracle, cfg, rows, S, n, (1, 7000, 33, n - 7034), 11)   # block cuts do not matter
    if cfg.n_channels == 3:
        assert dropped > 0, "three channels and lost channels: some note-ons must be dropped"


@pytest.mark.parametrize("index", range(9))
def test_birds_programs_match_oracle(api, oracle, index):
    # the reference's nine BIRDS programs (gen.voice.h:702-735): Markov-drawn ramp lists, pauses, cross-fading ramps, frequency transitions
    cfg, name = api.voice_program(api.VOICE_BIRDS, index, seed=1000 + index, n_channels=6)
    S, n = 2, 240000
    rows = note_rows(np.random.default_rng(100 + index), S, 7, (3000, 40000), (2000, 200000), freq=(150, 1200))
    check_against_oracle(api, oracle, cfg, rows, S, n, (5, 100000, 1, n - 100006), 1000 + index)


@pytest.mark.parametrize("index", range(2))
def test_robots_programs_match_oracle(api, oracle, index):
    cfg, name = api.voice_program(api.VOICE_ROBOTS, index, seed=2000 + index, n_channels=5)
    S, n = 2, 200000
    rows = note_rows(np.random.default_rng(200 + index), S, 8, (2000, 30000), (1500, 150000), freq=(150, 1500))
    check_against_oracle(api, oracle, cfg, rows, S, n, (70000, 1, 64, n - 70065), 2000 + index)


@pytest.mark.parametrize("kw", [dict(seed_param=48, markov_min_path_length=5, markov_additional_tries=20, xfade_freq=2, freq_transition_seconds=0.01, freq_interpolation=19,
                                     length_ms=12.0, length_exponent=1.0, articulative_pause_seconds=0.0, interpolation=4, freq_scatter=0.3, random_pan=0, pan=0.4),
                                dict(seed_param=0, markov_start_node=2, markov_pre_tries=3, markov_min_path_length=20, markov_additional_tries=20, length_ms=10.0,
                                     length_exponent=0.0, length_exponent_scatter=1.0, xfade_length_seconds=0.0454, articulative_pause_seconds=0.5, interpolation=22),
                                dict(sample_rate=44100, markov_min_path_length=4, xfade_freq=1, freq_transition_seconds=0.2, length_ms=30.0, length_exponent=2.0,
                                     interpolation=16, freq_interpolation=9, sine_gain=0.7, loudness_level=80.0, loudness_compensation_amount=0.3)])
def test_birds_variants(api, oracle, kw):
    # corners: the 30-spec cap with frequency transitions (cancel_last_ramp), ramps shorter than their cross-fade (skipped at once), every
    # ramp busy when the next spec is due, long pauses during which a note-off finds every ramp finished (the channel is not lost)
    cfg, _ = api.voice_program(api.VOICE_BIRDS, 1, seed=31, n_channels=4, **kw)
    S, n = 2, 160000
    rows = note_rows(np.random.default_rng(7), S, 10, (1000, 20000), (500, 100000), freq=(200, 900))
    check_against_oracle(api, oracle, cfg, rows, S, n, (n,), 31, expect_sound=(cfg.length_exponent > 0))


@pytest.mark.parametrize("index", range(14))
def test_wind_programs_match_oracle(api, oracle, index):
    # the reference's fourteen WIND programs (gen.voice.h:762-806): filter orders 1 .. 129, band-pass noise, the sine with and without short-term noise.
    # Notes held and released (a note-off silences the element at once), several at a time, blocks cut at odd places.
    cfg, name = api.voice_program(api.VOICE_WIND, index, seed=3000 + index, n_channels=4)
    S, n = 2, 60000
    rows = note_rows(np.random.default_rng(300 + index), S, 4, (500, 12000), (3000, 50000), freq=(150, 1200))
    check_against_oracle(api, oracle, cfg, rows, S, n, (3, 20000, 1, n - 20004), 3000 + index, min_rms=1e-9, rel_bar=1e-5)


@pytest.mark.parametrize("kw", [dict(pink_noise_br_gain=0.7, order_filters=3),                                   # the band-reject voice beside the band-pass
                                dict(pink_noise_bp_gain=0.0, pink_noise_br_gain=1.0, order_filters=70, bp_octave_width_min=0.5, bp_octave_width_max=2.0),
                                dict(sine_gain=0.2, center_short_term_ratio=0.6, seconds_slow_iter_short_term=-8.0, order_filters=258, interpolation=9),   # every voice, the largest order
                                dict(sample_rate=44100, sine_gain=0.3, pink_noise_br_gain=0.4, order_filters=64, seconds_slow_iter_long_term=-9.5, random_pan=0, pan=-0.3),
                                dict(sine_gain=0.5, pink_noise_bp_gain=0.0, center_short_term_ratio=1.0, seconds_slow_iter_short_term=-6.0, seconds_slow_iter_long_term=-10.5,
                                     interpolation=1, markov_start_node=1)])
def test_wind_variants(api, oracle, kw, monkeypatch):
    cfg, _ = api.voice_program(api.VOICE_WIND, 9, seed=77, n_channels=3, **kw)   # "Bubbles": order 129, widths and centres that move
    S, n = 2, 40000
    rows = note_rows(np.random.default_rng(17), S, 4, (200, 9000), (2000, 30000), freq=(200, 900))
    monkeypatch.setenv("RA_WIND_CHUNK", "4160")   # a ramp's iterators, filters and phase carried across chunks inside a block
    check_against_oracle(api, oracle, cfg, rows, S, n, (n // 2 + 1, n - n // 2 - 1), 77, min_rms=1e-9, rel_bar=1e-5)


def test_wind_many_notes_at_once(api, oracle):
    # forty WIND notes sounding together in one synth (48 channels; every note its own ramp, state slot and start positions), some released: the mix sums
    # the stream's ramps in job order
    cfg, _ = api.voice_program(api.VOICE_WIND, 13, seed=21, n_channels=48)   # "Kettle whistle mixed": band-pass of order 7 and the sine
    n = 24000
    rows = [(17 * k, 0, 0, k + 1, 150.0 + 23 * k, 0.05 + 0.01 * (k % 7)) for k in range(40)]
    rows += [(9000 + 31 * k, 0, 1, 3 * k + 1, 0.0, 0.0) for k in range(10)]
    rows.sort(key=lambda r: r[0])
    out, dropped, _ = check_against_oracle(api, oracle, cfg, rows, 1, n, (5000, n - 5000), 21, min_rms=1e-9, rel_bar=1e-5)
    assert dropped == 0


def test_wind_block_and_chunk_cuts_do_not_show(api, monkeypatch):
    # 64 synths x 3 held notes of "Bubbles" (order 129: five waves per band-pass chain) over one second: one block = ragged blocks = chunks of 4160 frames,
    # bit for bit (a noise voice's arithmetic does not depend on where a batch of 64 steps starts); synth s is not synth 0 (its own draws)
    cfg, _ = api.voice_program(api.VOICE_WIND, 9, seed=5, n_channels=4)
    S, n = 64, 48000
    rows = [(100 * k + 3 * s, s, 0, 3 * s + k + 1, 300.0 + 50 * k, 0.7) for s in range(S) for k in range(3)]
    rows.sort(key=lambda r: (r[1], r[0]))

    def run(cuts):
        with api.Voice(cfg, S, 0) as v:
            v.events(rows)
            return np.concatenate([v.process(m) for m in cuts], axis=1)
    whole = run((n,))
    assert np.sqrt(np.mean(whole.astype(np.float64) ** 2)) > 1e-5
    assert np.array_equal(whole, run((1, 777, 64, 20000, n - 20842)))
    monkeypatch.setenv("RA_WIND_CHUNK", "4160")
    assert np.array_equal(whole, run((n,)))
    assert not np.array_equal(whole[0], whole[1])


def test_wind_is_band_limited_noise(api):
    # "Steady wind": pink noise through an order-4 band-pass whose centre stays within 10 Hz * 2^5.2 .. 2^5.5 (368 .. 453 Hz), 1.3 octaves either side:
    # the spectrum peaks there and falls off on both sides; the level follows the attack of the cross-fade envelope
    cfg, _ = api.voice_program(api.VOICE_WIND, 1, random_pan=0, pan=0.0)
    with api.Voice(cfg, 1, 0) as v:
        v.events([(0, 0, 0, 1, 440.0, 1.0)])
        out = v.process(1 << 17)[0]
    x = out[1 << 14:, 0].astype(np.float64)
    assert np.sqrt(np.mean(x ** 2)) > 1e-3
    p = np.abs(np.fft.rfft(x * np.hanning(len(x)))) ** 2
    f = np.fft.rfftfreq(len(x), 1 / cfg.sample_rate)
    band = lambda lo, hi: p[(f >= lo) & (f < hi)].mean()
    mid = band(300, 550)
    assert mid > 20 * band(20, 40) and mid > 20 * band(5000, 8000)
    assert np.allclose(out[:, 0], out[:, 1])
    a = int(0.5 + 0.0499 * cfg.sample_rate)
    assert np.sqrt(np.mean(out[: a // 8, 0].astype(np.float64) ** 2)) < 0.2 * np.sqrt(np.mean(out[a:2 * a, 0].astype(np.float64) ** 2))


def test_many_channels_take_the_wide_workgroups(api, oracle):
    # more than 64 ramps of one stream at once: one (stream, tile) unit per workgroup of g > 64 lanes
    cfg, _ = api.voice_program(api.VOICE_ROBOTS, 1, seed=9, n_channels=120)
    n = 30000
    rows = [(10 + 3 * k, 0, 0, k + 1, 200.0 + 7 * k, 0.05) for k in range(110)]
    out, dropped, _ = check_against_oracle(api, oracle, cfg, rows, 1, n, (n,), 9)
    assert dropped == 0


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


def test_voice_is_bit_reproducible_and_hbm_resident_output_equals_host_output(api):
    # every output sample is one fixed-order sum (no atomics): two runs agree bit for bit, on the host path, on the device path and on the
    # asynchronous path over a caller's stream — with overlapping notes on every stream
    import torch
    cfg, _ = api.voice_program(api.VOICE_BIRDS, 8, seed=5, n_channels=8)
    rows = [(100 * k + 37 * s, s, 0, 10 * s + k, 200.0 + 50 * k, 0.8) for s in range(3) for k in range(5)]
    rows.sort(key=lambda r: (r[1], r[0]))
    n = 60000
    runs = []
    for _ in range(2):
        with api.Voice(cfg, 3, 0) as v:
            v.events(rows)
            runs.append(v.process(n))
    assert np.any(runs[0] != 0) and np.array_equal(runs[0], runs[1])
    with api.Voice(cfg, 3, 0) as v:
        v.events(rows)
        out = torch.empty(3, n, 2, dtype=torch.float32, device="cuda:0")
        v.process_device(n, out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), runs[0])
    st = torch.cuda.Stream()
    with api.Voice(cfg, 3, 0) as v:
        v.set_hip_stream(st.cuda_stream)
        v.events(rows)
        halves = [torch.empty(3, n // 2, 2, dtype=torch.float32, device="cuda:0") for _ in range(2)]
        for h in halves:
            v.process_device_async(n // 2, h.data_ptr())   # both blocks are enqueued before either is waited for
        v.synchronize()
        got = torch.cat(halves, dim=1).cpu().numpy()
    # (another block cut packs the ramps into other slots: the sums associate differently — a float ulp now and then)
    assert np.max(np.abs(got.astype(np.float64) - runs[0])) < 1e-6


def test_voice_six_blocks_in_flight_equal_blocks_waited_for_one_by_one(api):
    # the pre-pass of block k + 1 runs on the object's own queue beside the render of block k: the buffers a render reads exist twice, the
    # tile-boundary states three times, events order their reuse.  Six blocks of unequal lengths enqueued without a wait must give, bit for
    # bit, what the same six blocks give when each is waited for (same cuts, same slots, same sums).
    import torch
    cfg, _ = api.voice_program(api.VOICE_ROBOTS, 1, seed=11, n_channels=8)
    S = 24
    rows = [(2000 * k + 53 * s, s, 0, 100 * s + k, 180.0 + 35 * (k % 9), 0.7) for s in range(S) for k in range(40)]
    rows.sort(key=lambda r: (r[1], r[0]))
    cuts = [20000, 7000, 31000, 1000, 25000, 16000]
    def run(asynchronous):
        outs = [torch.empty(S, n, 2, dtype=torch.float32, device="cuda:0") for n in cuts]
        st = torch.cuda.Stream()
        with api.Voice(cfg, S, 0) as v:
            v.set_hip_stream(st.cuda_stream)
            v.events(rows)
            for n, o in zip(cuts, outs):
                if asynchronous:
                    v.process_device_async(n, o.data_ptr())
                else:
                    v.process_device(n, o.data_ptr())
            v.synchronize()
            dropped = v.dropped_note_ons()
        return torch.cat(outs, dim=1).cpu().numpy(), dropped
    a, da = run(True)
    b, db = run(False)
    assert np.any(a != 0) and da == db
    assert np.array_equal(a, b)


def test_voice_argument_errors(api):
    for kw in (dict(n_channels=0), dict(low_freq=0.0), dict(length_ms=0.0), dict(loudness_ref_freq_index=16), dict(length_ms=10.0, length_exponent=0.0, xfade_length_seconds=0.04),
               dict(mode=3, pink_noise_lp_gain=0.5), dict(mode=3, order_filters=0), dict(mode=3, order_filters=259), dict(mode=3, seconds_slow_iter_long_term=-20.0),
               dict(mode=4), dict(mode=7), dict(interpolation=23)):
        with pytest.raises(api.RaError) as e:
            api.Voice(api.voice_default_config(**kw), 1, 0)
        assert e.value.code == -1
    for kw in (dict(markov_start_node=3), dict(markov_additional_tries=21), dict(articulative_pause_seconds=0.6), dict(seed_param=258), dict(xfade_freq=1, freq_transition_seconds=0.0)):
        with pytest.raises(api.RaError):
            api.Voice(api.voice_program(api.VOICE_BIRDS, 0, **kw)[0], 1, 0)
    with pytest.raises(api.RaError):
        api.voice_program(api.VOICE_WIND, 14)
    with pytest.raises(api.RaError):
        api.Voice(api.voice_program(api.VOICE_WIND, 0, markov_start_node=2)[0], 1, 0)
    with api.Voice(api.voice_default_config(), 1, 0) as v:
        for bad in ([(0, 1, 0, 1, 440.0, 1.0)], [(0, 0, 2, 1, 440.0, 1.0)], [(0, 0, 0, 1, 0.0, 1.0)], [(0, 0, 0, 1, 440.0, 0.0)]):
            with pytest.raises(api.RaError):
                v.events(bad)
