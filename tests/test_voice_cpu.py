"""CPU tests of gen.voice's host half: the product's note expansion (cpp_audio_amd/csrc/ra_voice_host.hpp, through the
host-only C-ABI entry ra_voice_expand_note — no GPU needed) against the oracle's literal, per-sample SoundEngine; the restated MarkovChain
against its own transition table."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def api():
    from cpp_audio_amd import api
    api.lib()
    return api


def seeds_of(cfg, count):
    """the values setupAudioElement reseeds with, note after note (gen.voice.h:910-925)"""
    out, s = [], 0
    for _ in range(count):
        if cfg.mode != 2:
            s = cfg.seed_param if cfg.seed_param else (1 if s == 257 else s + 1)
        out.append(s)
    return out


def compare(api, oracle, cfg, stream_seed, freqs):
    trace = oracle.voice_trace(oracle.voice_params_from(cfg, stream_seed), freqs)
    for i, (f, sv, t) in enumerate(zip(freqs, seeds_of(cfg, len(freqs)), trace)):
        e = api.voice_expand_note(cfg, f, stream_seed, sv, noteid=i + 1)
        assert np.array_equal(e["specs_d"], t["specs_d"]), (i, "ramp specs: from / to / duration / start")
        assert np.array_equal(e["specs_i"], t["specs_i"]) and np.array_equal(e["vol"], t["vol"]), (i, "interpolation / silence / volume")
        assert np.array_equal(e["played"][:, :3], t["played"]), (i, "playNextSpec: element step, ramp, spec")
        assert e["done_step"] == t["done_step"], (i, "steps until the engine is finished")
        assert e["gain_left"] == t["gain_left"], (i, "pan")
    return trace


@pytest.mark.parametrize("mode, index", [(0, i) for i in range(9)] + [(1, 0), (1, 1), (2, 0), (2, 1)])
def test_note_expansion_equals_the_oracles_engine_for_the_reference_programs(api, oracle, mode, index):
    # the Markov chain's ramp specs bit for bit, and the schedule the product derives with integer arithmetic against what the oracle's
    # three ramp objects, stepped sample by sample through orchestrate_algos / playNextSpec, actually did
    cfg, name = api.voice_program(mode, index, seed=77)
    trace = compare(api, oracle, cfg, 77, [220.0, 330.0, 1000.0, 440.0, 95.0, 2500.0, 61.5, 880.0])
    if mode == 0 and index in (1, 2, 7, 8):
        assert any(np.any(t["specs_i"][:, 0] != cfg.interpolation) or len(t["specs_d"]) > 2 for t in trace)


@pytest.mark.parametrize("index", [0, 3, 11, 13])
def test_wind_note_expansion(api, oracle, index):
    # WIND (create_wind, soundengine.h:721-735): node 0's lambda takes a spec on ENTER and on LEAVE — one per step of the path — and sets the long-term
    # iterator's interpolation; playNextSpec plays spec 0 in the first element step, for 2^30 - xfade steps (the oracle's engine is stepped for
    # 2000 samples per note: the later plays, 2^30 steps apart, are the product's arithmetic alone)
    cfg, name = api.voice_program(3, index, seed=5)
    freqs = [220.0, 330.0, 1000.0]
    trace = oracle.voice_trace(oracle.voice_params_from(cfg, 5), freqs)
    xf = int(0.5 + cfg.xfade_length_seconds * cfg.sample_rate)
    for i, (f, sv, t) in enumerate(zip(freqs, seeds_of(cfg, len(freqs)), trace)):
        e = api.voice_expand_note(cfg, f, 5, sv, noteid=i + 1)
        assert len(t["specs_d"]) == 6 and np.array_equal(e["specs_d"], t["specs_d"]) and np.all(e["specs_d"][:, 2] == 2 ** 30 - 1)
        assert np.array_equal(e["specs_i"], t["specs_i"]) and np.all(e["specs_i"][:, 0] == cfg.interpolation)
        assert len(t["played"]) == 1 and np.array_equal(e["played"][0, :3], t["played"][0]) and tuple(t["played"][0]) == (1, 2, 0)
        assert e["played"][0, 3] == 2 ** 30 - xf and len(e["played"]) == 6 and np.all(np.diff(e["played"][:, 0]) == 2 ** 30 - xf)
        assert e["gain_left"] == t["gain_left"]


def test_wind_pink_noise_and_start_positions_equal_the_oracles(api, oracle):
    # the product's own restatement of the pink-noise sound buffer (ra_voice_host.hpp) against the oracle's (oracle_wind.hpp): every sample, the abs mean;
    # the declared start positions of a ramp's iterators
    for sr in (48000, 44100):
        a, am = api.voice_pink_noise(sr)
        b, bm = oracle.pink_noise(sr)
        assert len(a) == int(np.float32(sr) / np.float32(0.05)) and np.array_equal(a, b) and am == bm
    rng = np.random.default_rng(4)
    for _ in range(200):
        key, noteid, play, it, which = int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 40)), int(rng.integers(0, 30)), int(rng.integers(0, 6)), int(rng.integers(0, 2))
        assert api.voice_wind_start_position(key, noteid, play, it, which, 960000) == oracle.wind_start_position(key, noteid, play, it, which, 960000)


def test_note_expansion_over_random_programs(api, oracle):
    rng = np.random.default_rng(2024)
    long_lists = skipped = waits = 0
    for case in range(60):
        mode = int(rng.integers(0, 2))
        cfg, _ = api.voice_program(mode, 0, seed=int(rng.integers(1, 1 << 40)))
        cfg.sample_rate = int(rng.choice([22050, 44100, 48000, 96000]))
        cfg.interpolation = int(rng.integers(0, 23))
        cfg.markov_start_node = int(rng.integers(0, 3))
        cfg.markov_pre_tries = int(rng.integers(0, 21))
        cfg.markov_min_path_length = int(rng.integers(0, 21))
        cfg.markov_additional_tries = int(rng.integers(0, 21))
        cfg.articulative_pause_seconds = float(rng.choice([0.0, 0.001, 0.02, 0.5]))
        cfg.freq_scatter = float(rng.uniform(0, 1))
        cfg.length_ms = float(rng.uniform(10, 60))
        cfg.length_exponent = float(rng.uniform(0, 3))
        cfg.length_exponent_scatter = float(rng.uniform(0, 1))
        cfg.xfade_length_seconds = float(rng.uniform(0.0023, 0.0454))
        cfg.phase_ratio1 = float(rng.uniform(-0.2, 1.2))
        cfg.seed_param = int(rng.choice([0, 0, 1, 257, 48]))
        cfg.random_pan = int(rng.integers(0, 2))
        cfg.pan = float(rng.uniform(-1, 1))
        if mode == 0:
            cfg.xfade_freq = int(rng.integers(0, 3))
            cfg.freq_transition_seconds = float(rng.uniform(0.001, 0.5))
            cfg.freq_interpolation = int(rng.integers(0, 23))
        else:
            cfg.d1, cfg.d2 = float(rng.integers(0, 48)), float(rng.integers(0, 48))
            cfg.harmonic_attenuation = float(rng.uniform(0.5, 1.1))
        trace = compare(api, oracle, cfg, cfg.seed + 3, [float(f) for f in rng.uniform(40, 4000, 4)])
        for t in trace:
            long_lists += len(t["specs_d"]) == 30
            xf = int(0.5 + cfg.xfade_length_seconds * cfg.sample_rate)
            T = np.maximum((0.5 + t["specs_d"][t["played"][:, 2], 2]).astype(np.int64) - xf, 0) if len(t["played"]) else np.zeros(0)
            skipped += int(np.sum(T == 0))
            if len(t["played"]) > 3:
                k = t["played"][:, 0]
                waits += int(np.any(np.diff(k) > T[:-1] + int(cfg.articulative_pause_seconds * cfg.sample_rate)))
    # the corners this sweep is meant to reach: the 30-spec cap, ramps shorter than their cross-fade, a spec that had to wait for a free ramp
    assert long_lists > 0 and skipped > 0 and waits > 0


def transition_matrix(which, normalized):
    edges = {0: [(0, 1, 0.5), (1, 0, 0.015), (0, 2, 0.5), (2, 0, 0.015), (2, 1, 0.885)],                       # create_birds, soundengine.h:588-592
             1: [(0, 1, 1.0), (1, 2, 0.2), (2, 1, 0.1), (1, 3, 0.2), (3, 1, 1.0)]}[which]                     # create_robot, :688-692
    n = 3 if which == 0 else 4
    P = np.zeros((n, n))
    for a, b, p in edges:
        P[a, b] += np.float32(p)
    for a in range(n):
        s = P[a].sum()
        if normalized and s > 0:
            P[a] /= s
        else:
            P[a, a] += 1.0 - s        # a raw step stays where it is with the remaining probability
    return P


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("normalized", [False, True])
def test_markov_chain_visits_follow_the_transition_table(oracle, which, normalized):
    # the restated MarkovChain driven by the declared draws: its visit frequencies over 2 M steps are those of the table's stationary law
    P = transition_matrix(which, normalized)
    n = P.shape[0]
    steps = 2_000_000
    visits = oracle.markov_visits(which, 0, normalized, 12345, steps)[:n].astype(np.float64)
    w, v = np.linalg.eig(P.T)
    pi = np.real(v[:, np.argmin(np.abs(w - 1.0))])
    pi = pi / pi.sum()
    # slow mixing (the birds chain leaves node 1 once in 67 steps): allow for the autocorrelation, 6 sigma of an effective sample of steps / 200
    tol = 6 * np.sqrt(pi * (1 - pi) / (steps / 200)) + 1e-9
    assert np.all(np.abs(visits / steps - pi) < tol), (visits / steps, pi)
    assert np.all((pi > 0) == (visits > 0))   # the robot's node 0 is left for good


def test_expand_note_argument_errors(api):
    cfg, _ = api.voice_program(0, 0)
    with pytest.raises(api.RaError):
        api.voice_expand_note(cfg, 0.0, 1, 1)
    cfg.mode = 4
    with pytest.raises(api.RaError) as e:
        api.voice_expand_note(cfg, 440.0, 1, 1)
    assert "mode" in str(e.value)
    cfg, _ = api.voice_program(3, 0, pink_noise_lp_gain=0.25)
    with pytest.raises(api.RaError) as e:
        api.voice_expand_note(cfg, 440.0, 1, 1)
    assert "pink_noise_lp_gain" in str(e.value)
    cfg, _ = api.voice_program(1, 1)
    with pytest.raises(api.RaError) as e:
        api.voice_expand_note(cfg, 440.0, 1, 1, spec_cap=2)
    assert e.value.code == -4
