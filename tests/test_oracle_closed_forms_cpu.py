"""CPU checks of the oracle against closed forms (the reference holds no vectors for these, so the restatement is held against
independent mathematics instead): the sine bank element against A env(n) sin(phi0 + pi sum(inc)), the AHDSR against its piecewise-linear law
with the minimum-change floor, the frequency glide against its exponential law, gen.voice's LogRamp against the sweep it is built to follow,
the carrier's saw / triangle / pulse against the sample index of every edge, the limiter against its ceiling and recovery constant."""
import numpy as np
import pytest

SR = 48000


def ahdsr_closed_form(n, K, A, H, D, R, S):
    """Envelope value after step j = 1 .. n of a note pressed before step 1 and released before step K + 1 (audioelement.h:702-756, 960-1000):
    linear 0 -> 1 over A steps, 1 for H, linear 1 -> S over D, S; then linear top -> 0 over R steps from the value at the release."""
    j = np.arange(1, n + 1, dtype=np.float64)
    pressed = np.where(j <= A, j / A, np.where(j <= A + H, 1.0, np.where(j <= A + H + D, 1.0 + (S - 1.0) * (j - A - H) / max(D, 1), S)))
    top = pressed[K - 1]
    m = j - K
    return np.where(j <= K, pressed, np.where(m < R, top - top * m / R, 0.0))


@pytest.mark.parametrize("freq, ahdsr, sustain", [(440.0, (100, 50, 200, 300), 0.6), (55.0, (10, 0, 0, 10), 1.0), (3000.0, (480, 0, 960, 2400), 0.25),
                                                  (12000.0, (0, 0, 0, 0), 1.0)])
def test_sine_element_against_its_closed_form(oracle, freq, ahdsr, sustain):
    # steady note: out(j) = gain * (volume * (alias * sin(pi (phi0 + j inc)) * env(j)))   
    inc = float(np.float32(2 * freq / SR))
    vol, phi0, pan = 0.37, 0.3, -0.4
    K, n = 6000, 12000
    out = oracle.elem_run([(0, 2, pan), (0, 1, vol), (0, 0, inc), (0, 3, phi0), (0, 4, 0), (K, 5, 0)], n, ahdsr=ahdsr, sustain=sustain)
    min_dt, min_change = SR // 1000, int(0.5 + 2.5 * 2 / inc)      # audioelement.h:863-872, 216-223
    has_decay = sustain < 0.999999
    A, H = max(ahdsr[0], min_dt, min_change), ahdsr[1]
    D = max(ahdsr[2], min_dt, min_change) if has_decay else 0
    R = max(ahdsr[3], min_dt, min_change)
    S = float(np.float32(sustain)) if has_decay else 1.0
    env = ahdsr_closed_form(n, K, A, H, D, R, S)
    assert np.max(np.abs(out[:, 1] - env)) < 1e-15
    half_period = 1.0 / inc
    alias = 1.0 if half_period > 4 else max(0.0, (half_period - 1.0) / 3.0)     # audioelement.h:466-483
    j = np.arange(1, n + 1)
    gain = np.cos(np.pi / 4 * (pan + 1))
    expect = gain * (vol * (alias * np.sin(np.pi * (phi0 + inc * j)) * env))
    assert np.max(np.abs(expect)) > 0.05
    assert np.max(np.abs(out[:, 0] - expect)) < 2e-13                            # the phasor recurrence against the closed-form phase
    assert np.all(out[:K + R - 1, 3] == vol)                                     # the volume filter starts at its target and stays
    # RT-active until the release has run out and EnvelopeDone1 has lasted 17 steps
    assert out[K + R + 15, 5] == 1.0 and out[K + R + 16, 5] == 0.0


def test_release_during_the_attack_starts_from_the_value_reached(oracle):
    inc = float(np.float32(2 * 200.0 / SR))
    out = oracle.elem_run([(0, 2, 0.0), (0, 1, 1.0), (0, 0, inc), (0, 3, 0.0), (0, 4, 0), (300, 5, 0)], 3000, ahdsr=(2000, 0, 0, 1000), sustain=1.0)
    min_change = int(0.5 + 2.5 * 2 / inc)
    A, R = max(2000, min_change), max(1000, min_change)
    env = ahdsr_closed_form(3000, 300, A, 0, 0, R, 1.0)
    assert abs(env[299] - 300 / A) < 1e-15 and np.max(np.abs(out[:, 1] - env)) < 1e-15


@pytest.mark.parametrize("f0, f1", [(220.0, 440.0), (1000.0, 250.0), (3000.0, 3100.0)])
def test_glide_follows_the_exponential_law_and_the_phase_is_the_sum_of_the_increments(oracle, f0, f1):
    # InterpolatedFreq (audioelement.h:2764-2816): after setAngleIncrements(to) the increment goes from -> to over `stride` samples with its
    # time warped in proportion to the current frequency, i.e. the frequency is exponential in real time: inc(k) = from (to / from)^(k / stride)
    stride = 1024
    a, b = float(np.float32(2 * f0 / SR)), float(np.float32(2 * f1 / SR))
    K = 4000
    n = K + 3 * stride
    out = oracle.elem_run([(0, 2, 0.0), (0, 1, 1.0), (0, 0, a), (0, 3, 0.0), (0, 4, 0), (K, 0, b)], n, ahdsr=(48, 0, 0, 48), sustain=1.0, stride=stride)
    inc = out[:, 2]
    assert np.all(inc[:K] == a)
    k = np.arange(stride)
    law = a * (b / a) ** (k / stride)
    got = inc[K:K + stride]
    assert np.max(np.abs(got / law - 1.0)) < 1.5 * abs(np.log(b / a)) / stride      # one step of the law's own granularity
    assert np.all(inc[K + stride + 2:] == b) and np.any(inc[K + stride - 3:K + stride] != b)   # it arrives after `stride` samples, to within rounding of the warp
    assert np.all(np.diff(got) * np.sign(b - a) > 0)
    # whatever the increments are, the oscillator's phase is their running sum (SineOscillatorAlgo::step, :2413-2421)
    phase = np.cumsum(inc)
    unit = np.exp(1j * np.pi * (out[:, 4] - phase))
    assert np.max(np.abs(unit - 1.0)) < 1e-11
    amp = out[:, 0] / (np.cos(np.pi / 4) * out[:, 1] * out[:, 3])
    m = out[:, 1] == 1.0
    assert np.max(np.abs(amp[m] - np.sin(np.pi * phase[m]))) < 1e-11


@pytest.mark.parametrize("lo, hi, dur", [(100.0, 20000.0, 112128.0), (8000.0, 60.0, 15360.0), (300.0, 310.0, 5000.0)])
def test_logramp_is_an_exponential_sweep_of_its_duration(oracle, lo, hi, dur):
    # LogRamp (audioelement.h:2464-2572): "the same effect as PROPORTIONAL_VALUE_DERIVATIVE": the frequency is exponential in time and
    # reaches `to` after duration_in_samples steps; then the ramp swaps its ends and comes back (periodic)
    a, b = float(np.float32(2 * lo / SR)), float(np.float32(2 * hi / SR))
    n = int(2.2 * dur)
    inc = oracle.logramp_run(a, b, dur, 0.0, 0, n)
    turn = int(np.argmax(inc)) if b > a else int(np.argmin(inc))
    assert abs(turn - dur) <= 2 + 2e-4 * dur                                       # the sweep's duration (the warp is an Euler scheme: 16 of 112 128 steps early at 200 : 1)
    k = np.arange(turn)
    law = a * (b / a) ** (k / dur)
    step = abs(np.log(b / a)) / dur                                                # the law's relative change per sample
    assert np.max(np.abs(inc[:turn] / law - 1.0)) < 20 * step + 1e-12             # never more than a few samples of the law away
    back = inc[turn + 1:turn + 1 + turn // 2]
    assert np.max(np.abs(back / (b * (a / b) ** (np.arange(len(back)) / dur)) - 1.0)) < 40 * step + 1e-12  # and comes back the same way
    # a constant ramp keeps its value and its time runs at `from` per step (the 1.f of :2570): it never turns
    c = oracle.logramp_run(a, a, 1000.0, 0.0, 0, 5000)
    assert np.all(c == a)


def test_easing_functions_have_their_published_shapes(oracle):
    # Penner's equations as decided for itp::interpolate: end points, monotony, in / out mirror symmetry, in-out point symmetry
    t = np.linspace(0.0, 7.0, 1401)
    b, c, d = 0.25, 1.5, 7.0
    names = ["QUAD", "CUBIC", "QUART", "QUINT", "SINE", "EXPO", "CIRC"]
    for fam in range(7):
        k_in, k_out, k_io = 2 + 3 * fam, 3 + 3 * fam, 4 + 3 * fam
        f_in, f_out, f_io = (oracle.itp(k, t, b, c, d) for k in (k_in, k_out, k_io))
        tol = 2e-3 if names[fam] == "EXPO" else 1e-12                              # Penner's expo curves start / end 2^-10 off their end points
        for f in (f_in, f_out, f_io):
            assert abs(f[0] - b) <= tol * c and abs(f[-1] - (b + c)) <= tol * c, names[fam]
            assert np.all(np.diff(f) >= -1e-12), names[fam]
        assert np.max(np.abs(f_out - (2 * b + c - f_in[::-1]))) <= 2 * tol * c, names[fam]    # out(t) = b + c - (in(d - t) - b)
        assert np.max(np.abs(f_io + f_io[::-1] - (2 * b + c))) <= 2 * tol * c, names[fam]     # in-out is symmetric about the centre
        assert f_in[700] < b + c / 2 < f_out[700], names[fam]
    assert np.max(np.abs(oracle.itp(0, t, b, c, d) - (b + c * t / d))) < 1e-15
    pvd = oracle.itp(1, t, b, c, d)
    assert np.max(np.abs(pvd - b * ((b + c) / b) ** (t / d))) < 1e-14 and np.all(np.diff(np.log(pvd)) > 0)


def carrier_voice(oracle, voice, freq, n, pulse_width=None):
    kw = {f"vocoder_carrier_{v}_volume": (1.0 if v == voice else 0.0) for v in ("saw", "triangle", "square", "sine", "pulse")}
    if pulse_width is not None:
        kw["vocoder_carrier_pulse_width"] = pulse_width
    cfg = oracle.default_config(carrier_n_channels=2, carrier_volume=1.0, voice_volume=0.0, vocoder_volume=0.0, analysis_volume=0.0, phase_policy=1, **kw)
    ev = oracle.carrier_events([(0, 0, 69, 0, freq, 1.0)])
    _, car, _, dr = oracle.run_offline_ex(cfg, None, None, ev, n_frames=n, want_carrier=True)
    assert dr == (0, 0)
    return car


def phased_angle(inc, n):
    """Phased::step (audioelement.h:1395-1477): angle += inc; above 2 it comes back by 2 — as exact rationals of the float32 increment."""
    from fractions import Fraction
    a, two, step = Fraction(0), Fraction(2), Fraction(float(inc))
    out = np.empty(n)
    wraps = []
    for i in range(n):
        a += step
        if a > two:
            a -= two
            wraps.append(i)
        out[i] = float(a)
    return out, wraps


@pytest.mark.parametrize("freq", [110.0, 997.0])
def test_carrier_edges_fall_on_the_samples_of_the_exact_angle(oracle, freq):
    # 10 s of the discontinuous voices: the sample index of every edge equals that of an exact-arithmetic angle (no drift of the double
    # accumulation over 480 000 steps), and between the edges the waves are their closed forms (sound.functions.h:86-138)
    n = 10 * SR
    inc = np.float32(2) * np.float32(freq) / np.float32(SR)
    ang, wraps = phased_angle(inc, n)
    A = max(SR // 1000, int(0.5 + 2.5 * (2.0 / float(inc))))
    k = np.arange(n)
    env = np.clip((k + 1) / A, 0.0, 1.0)
    assert abs(len(wraps) - 10 * freq) <= 1
    saw = carrier_voice(oracle, "saw", freq, n)
    expect = np.where(ang <= 1.0, ang, ang - 2.0) * env
    assert np.max(np.abs(saw - expect)) < 1e-9                                     # the double angle is within 1e-10 of the exact one after 10 s
    edges = np.nonzero(np.diff(saw[A:]) < -0.5)[0] + A + 1                          # the saw's fall from +1 to -1 (angle passing 1)
    exact = np.nonzero((ang[A:-1] <= 1.0) & (ang[A + 1:] > 1.0))[0] + A + 1
    assert np.array_equal(edges, exact) and len(edges) > 1000
    tri = carrier_voice(oracle, "triangle", freq, n)
    expect = np.where(ang < 0.5, 2 * ang, np.where(ang < 1.5, 2.0 - 2.0 * ang, 2.0 * ang - 4.0)) * env
    assert np.max(np.abs(tri - expect)) < 2e-9
    w = 0.3
    pul = carrier_voice(oracle, "pulse", freq, n, pulse_width=w)
    wf = float(np.float32(w))
    high = 0.5 * (2.0 - wf)
    expect = np.where(ang < wf, high, high - 1.0) * env                             # PulseTrainAlgo_: mean-free pulse train (audioelement.h:1691-1748)
    assert np.array_equal(np.nonzero(np.diff(np.sign(pul[A:])))[0], np.nonzero(np.diff(np.sign(expect[A:])))[0])
    assert np.max(np.abs(pul - expect)) < 1e-12


def test_limiter_keeps_its_ceiling_and_recovers_with_its_time_constant(oracle):
    # the Limiter decided in oracle_dsp.hpp (cpp.algorithms' is absent): ceiling 0.95 on the target at once, the applied gain follows by 1 % per
    # frame, the target recovers by 2e-5 of the remaining distance per frame
    rng = np.random.default_rng(3)
    n = 400000
    x = np.zeros((n, 2))
    x[:, 0] = 0.2 * np.sin(2 * np.pi * 300 * np.arange(n) / SR)
    x[:, 1] = 0.2 * np.cos(2 * np.pi * 300 * np.arange(n) / SR)
    x[50000:60000] *= 12.0                                                        # a burst far above the ceiling
    x[200000:200010, 0] = 3.0 * rng.uniform(0.9, 1.0, 10)
    y, tgt = oracle.limiter_run(x)
    assert np.all(np.abs(y) <= 1.0)
    peak_in = np.max(np.abs(x), axis=1)
    assert np.all(tgt * peak_in <= 0.95 * (1 + 1e-12))                            # the target never asks for more than the ceiling allows
    steady = slice(52000, 60000)                                                  # the applied gain has caught up with the target (0.99^2000 ~ 2e-9)
    assert 0.94 < np.max(np.abs(y[steady])) <= 0.952                              # (between two peaks of the burst the target recovers a little)
    quiet = y[:50000]
    assert np.array_equal(quiet, x[:50000])                                       # below the ceiling the limiter is transparent
    # recovery: 1 - target decays by (1 - 2e-5) per frame after the burst
    d = 1.0 - tgt[60000:199999]
    rate = d[1:] / d[:-1]
    assert np.max(np.abs(rate - (1.0 - 2e-5))) < 1e-12
    # attack of the applied gain: 1 % of the distance per frame
    g = np.sum(y[50000:50400] * x[50000:50400], axis=1) / np.sum(x[50000:50400] ** 2, axis=1)     # the one gain both channels share
    t0 = tgt[50000:50400]
    free = np.max(np.abs(y[50000:50400]), axis=1) < 1.0                           # (while the gain is still too high the product is clipped to +-1)
    ok = free[1:] & free[:-1]
    assert ok.sum() > 100 and np.max(np.abs(np.diff(g) - 0.01 * (t0[1:] - g[:-1]))[ok]) < 1e-9
