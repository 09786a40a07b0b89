"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class OraConfig(C.Structure):
    """Mirror of oracle/oracle_config.h:ora_config."""
    _fields_ = [
        ("sample_rate", C.c_int32), ("window_size", C.c_int32), ("stride", C.c_int32),
        ("n_channels", C.c_int32), ("window_kind", C.c_int32),
        ("min_volume", C.c_float), ("nearby_distance_tones", C.c_float), ("max_track_pitches", C.c_float),
        ("autotune_tolerance_pitches", C.c_float),
        ("pitch_shift_pre_autotune", C.c_float), ("pitch_shift_post_autotune", C.c_float),
        ("pitch_harmonize_pre_autotune", C.c_float), ("pitch_harmonize_post_autotune", C.c_float),
        ("stereo_spread", C.c_float),
        ("env_attack_seconds", C.c_float), ("env_hold_seconds", C.c_float), ("env_decay_seconds", C.c_float),
        ("env_release_seconds", C.c_float), ("env_sustain_level", C.c_float),
        ("use_autotune", C.c_int32), ("autotune_type", C.c_int32), ("autotune_max_pitch", C.c_int32),
        ("autotune_factor", C.c_int32), ("autotune_scale_mode", C.c_int32), ("autotune_root_note", C.c_int32),
        ("autotune_root_transpose", C.c_int32), ("autotune_chord_frequencies", C.c_int32),
        ("autotune_bit_chord", C.c_uint64),
        ("vocoder_modulator_window_size_seconds", C.c_float), ("vocoder_stride_seconds", C.c_float),
        ("vocoder_count_bands", C.c_int32), ("vocoder_min_freq", C.c_float), ("vocoder_max_freq", C.c_float),
        ("voice_volume", C.c_float), ("carrier_volume", C.c_float), ("vocoder_volume", C.c_float),
        ("analysis_volume", C.c_float), ("analysis_input_gain", C.c_float),
        ("phase_policy", C.c_int32), ("seed", C.c_uint64),
        ("analysis_output_feedback_gain", C.c_float), ("output_delay_seconds", C.c_float),
        ("vocoder_kind", C.c_int32), ("vocoder_env_follower_cutoff_ratio", C.c_float), ("carrier_n_channels", C.c_int32),
        ("vocoder_carrier_noise_volume", C.c_float), ("vocoder_carrier_saw_volume", C.c_float),
        ("vocoder_carrier_triangle_volume", C.c_float), ("vocoder_carrier_square_volume", C.c_float),
        ("vocoder_carrier_sine_volume", C.c_float), ("vocoder_carrier_pulse_volume", C.c_float),
        ("vocoder_carrier_pulse_width", C.c_float), ("post", C.c_int32),
    ]


CARRIER_EVENT_DTYPE = np.dtype([("sample", "<i8"), ("type", "<i4"), ("key", "<i4"), ("delay", "<i4"), ("frequency", "<f4"),
                                ("velocity", "<f4"), ("_pad", "<i4")])  # oracle_config.h: ora_carrier_event (32 bytes)


def carrier_events(rows):
    """rows: iterable of (sample, type, key[, delay[, frequency[, velocity]]]) -> structured array sorted by sample (stable)."""
    ev = np.zeros(len(rows), CARRIER_EVENT_DTYPE)
    for i, r in enumerate(rows):
        r = tuple(r) + (0, 0.0, 1.0)[len(r) - 3:] if len(r) < 6 else tuple(r)
        ev[i] = (r[0], r[1], r[2], r[3], r[4], r[5], 0)
    return ev[np.argsort(ev["sample"], kind="stable")]


class OraEvent(C.Structure):
    _fields_ = [("sample", C.c_int64), ("type", C.c_int32), ("channel", C.c_int32), ("noteid", C.c_int64),
                ("frequency", C.c_float), ("velocity", C.c_float)]


EVENT_DTYPE = np.dtype([("sample", "<i8"), ("type", "<i4"), ("channel", "<i4"), ("noteid", "<i8"),
                        ("frequency", "<f4"), ("velocity", "<f4")])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.ora_midi_get_pitch.restype = C.c_double
        L.ora_scale_closest_pitch.restype = C.c_double
        L.ora_scale_closest_pitch.argtypes = [C.c_int, C.c_double, C.c_double]
        L.ora_frequency_to_midi_pitch.restype = C.c_double
        L.ora_frequency_to_midi_pitch.argtypes = [C.c_double]
        L.ora_midi_pitch_to_freq.restype = C.c_double
        L.ora_det_math.restype = None
        L.ora_det_math.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ora_midi_pitch_to_freq.argtypes = [C.c_double]
        L.ora_equal_loudness_volume_db.restype = C.c_float
        L.ora_equal_loudness_volume_db.argtypes = [C.c_double, C.c_int]
        L.ora_stereo.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ora_hash_uniform_pm1.restype = C.c_float
        L.ora_hash_uniform_pm1.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_autotune.restype = C.c_double
        L.ora_autotune.argtypes = [C.POINTER(OraConfig), C.c_double]
        L.ora_run_offline.restype = C.c_int64
        L.ora_run_frames.restype = C.c_int64
        L.ora_resynth_notes.restype = C.c_int64
        L.ora_peaks.argtypes = [C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_note_add_halftones.argtypes = [C.c_int, C.c_int, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ora_even_window_size_frames.argtypes = [C.c_float, C.c_int]
        L.ora_window_center_stride_frames.argtypes = [C.c_float, C.c_int]
        L.ora_good_stride.argtypes = [C.c_float, C.c_int]
        L.ora_modulator_window_size.argtypes = [C.c_float, C.c_int]
        _lib = L
    return _lib


class libm_mode:
    """with oracle_lib.libm_mode(): ... — the oracle evaluates log2 / log10 / pow with the platform libm, as the reference's text does
    (independent of the product's explicit series); outside, the explicit series (bit-comparable events)."""

    def __enter__(self):
        lib().ora_set_libm_mode(1)

    def __exit__(self, *a):
        lib().ora_set_libm_mode(0)


def default_config(**kw):
    c = OraConfig()
    lib().ora_default_config(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def white_noise(sample_rate):
    """The white-noise sound buffer of the carrier's noise voice (getWhiteNoise, source/sound.cpp:3-21; draws: the declared policy)."""
    L = lib()
    L.ora_white_noise.argtypes = [C.c_int32, C.c_void_p, C.c_int64]
    L.ora_white_noise.restype = C.c_int64
    n = L.ora_white_noise(int(sample_rate), None, 0)
    out = np.zeros(n, np.float64)
    L.ora_white_noise(int(sample_rate), _p(out), n)
    return out


def half_window(kind, n):
    out = np.empty(n, np.float64)
    lib().ora_half_window(C.c_int(kind), C.c_int(n), _p(out))
    return out


def real_fft(x):
    x = np.ascontiguousarray(x, np.float64)
    L = len(x)
    out = np.empty(2 * (L // 2 + 1), np.float64)
    lib().ora_real_fft(_p(x), C.c_int(L), _p(out))
    return out[0::2] + 1j * out[1::2]


def real_ifft(X, L):
    b = np.empty(2 * (L // 2 + 1), np.float64)
    b[0::2] = X.real
    b[1::2] = X.imag
    x = np.empty(L, np.float64)
    lib().ora_real_ifft(_p(b), C.c_int(L), _p(x))
    return x


def sqmag(x, window_kind=0):
    x = np.ascontiguousarray(x, np.float64)
    n = len(x)
    L = lib().ora_fft_length_for(C.c_int(n))
    out = np.empty(L // 2, np.float64)
    L2 = lib().ora_sqmag(_p(x), C.c_int(n), C.c_int(window_kind), _p(out))
    assert L2 == L
    return out, L


def peaks(sr, sq, L):
    sq = np.ascontiguousarray(sq, np.float64)
    cap = len(sq)
    f = np.empty(cap, np.float64)
    d = np.empty(cap, np.float64)
    n = lib().ora_peaks(C.c_double(sr), _p(sq), C.c_int(len(sq)), C.c_int(L), _p(f), _p(d), C.c_int(cap))
    return f[:n], d[:n]


def step_stateless(cfg, freq, db):
    freq = np.ascontiguousarray(freq, np.float64)
    db = np.ascontiguousarray(db, np.float64)
    cap = 2 * len(freq) + 8
    p = np.empty(cap, np.float64)
    v = np.empty(cap, np.float64)
    k = np.empty(cap, np.float32)
    n = lib().ora_step_stateless(C.byref(cfg), _p(freq), _p(db), C.c_int(len(freq)), _p(p), _p(v), _p(k), C.c_int(cap))
    return p[:n], v[:n], k[:n]


def run_offline(cfg, voice, carrier=None, log_cap=1 << 20):
    """Returns (out[n,2] float32, events structured array, dropped_note_on)."""
    n = len(voice) if voice is not None else len(carrier)
    v = np.ascontiguousarray(voice, np.float32) if voice is not None else None
    c = np.ascontiguousarray(carrier, np.float32) if carrier is not None else None
    out = np.zeros((n, 2), np.float32)
    log = np.zeros(log_cap, EVENT_DTYPE)
    dropped = C.c_int32(0)
    ne = lib().ora_run_offline(C.byref(cfg), _p(v) if v is not None else None, _p(c) if c is not None else None,
                               C.c_int64(n), _p(out), _p(log), C.c_int64(log_cap), C.byref(dropped))
    assert ne <= log_cap, "event log overflow"
    return out, log[:ne], dropped.value


def run_offline_ex(cfg, voice, carrier=None, events=None, n_frames=None, want_carrier=False, log_cap=1 << 20):
    """run_offline with carrier-synth note events.  Returns (out[n,2] float32, carrier double[n] or None, event log, (dropped, carrier dropped))."""
    n = n_frames if n_frames is not None else (len(voice) if voice is not None else len(carrier))
    v = np.ascontiguousarray(voice, np.float32) if voice is not None else None
    c = np.ascontiguousarray(carrier, np.float32) if carrier is not None else None
    ev = events if events is not None else np.zeros(0, CARRIER_EVENT_DTYPE)
    assert ev.dtype == CARRIER_EVENT_DTYPE
    out = np.zeros((n, 2), np.float32)
    car = np.zeros(n, np.float64) if want_carrier else None
    log = np.zeros(log_cap, EVENT_DTYPE)
    dropped = (C.c_int32 * 2)()
    L = lib()
    L.ora_run_offline_ex.restype = C.c_int64
    ne = L.ora_run_offline_ex(C.byref(cfg), _p(v) if v is not None else None, _p(c) if c is not None else None, C.c_int64(n),
                              _p(ev) if len(ev) else None, C.c_int64(len(ev)), _p(out), _p(car) if car is not None else None,
                              _p(log), C.c_int64(log_cap), dropped)
    assert ne <= log_cap
    return out, car, log[:ne], (dropped[0], dropped[1])


def run_frames(cfg, voice, frame_cap=1 << 16, item_cap=1 << 22):
    v = np.ascontiguousarray(voice, np.float32)
    fire = np.zeros(frame_cap, np.int64)
    offs = np.zeros(frame_cap + 1, np.int64)
    pitch = np.zeros(item_cap, np.float64)
    vol = np.zeros(item_cap, np.float64)
    nf = lib().ora_run_frames(C.byref(cfg), _p(v), C.c_int64(len(v)), _p(fire), _p(offs), C.c_int64(frame_cap),
                              _p(pitch), _p(vol), C.c_int64(item_cap))
    assert nf <= frame_cap and offs[nf] <= item_cap
    return fire[:nf], offs[:nf + 1], pitch[:offs[nf]], vol[:offs[nf]]


def run_offline_batch(cfg, voice, carrier=None, n_threads=1):
    """voice: [streams, n] float32. Returns out [streams, n, 2]."""
    a = voice if voice is not None else carrier
    S, n = a.shape
    v = np.ascontiguousarray(voice, np.float32) if voice is not None else None
    c = np.ascontiguousarray(carrier, np.float32) if carrier is not None else None
    out = np.zeros((S, n, 2), np.float32)
    lib().ora_run_offline_batch(C.byref(cfg), _p(v) if v is not None else None, _p(c) if c is not None else None,
                                C.c_int32(S), C.c_int64(n), _p(out), C.c_int32(n_threads))
    return out


def resynth_notes(start, end, freq, amp_db, stride, sr, cap=1 << 24):
    start = np.ascontiguousarray(start, np.int32)
    end = np.ascontiguousarray(end, np.int32)
    freq = np.ascontiguousarray(freq, np.float64)
    amp_db = np.ascontiguousarray(amp_db, np.float64)
    out = np.zeros(cap, np.float64)
    n = lib().ora_resynth_notes(_p(start), _p(end), _p(freq), _p(amp_db), C.c_int32(len(start)), C.c_int32(stride),
                                C.c_int32(sr), _p(out), C.c_int64(cap))
    assert n <= cap
    return out[:n]


def deduce_notes(x, sr, half_window_size, stride, window_kind=0, cap=1 << 20):
    """(start, end, freq, amp_db) arrays of the notes the oracle deduces from the mono float32 signal x."""
    x = np.ascontiguousarray(x, np.float32)
    st = np.zeros(cap, np.int32)
    en = np.zeros(cap, np.int32)
    fr = np.zeros(cap, np.float64)
    am = np.zeros(cap, np.float64)
    L = lib()
    L.ora_deduce_notes.restype = C.c_int64
    n = L.ora_deduce_notes(_p(x), C.c_int64(len(x)), C.c_int32(sr), C.c_int32(half_window_size), C.c_int32(window_kind), C.c_int32(stride),
                           _p(st), _p(en), _p(fr), _p(am), C.c_int64(cap))
    assert n <= cap
    return st[:n], en[:n], fr[:n], am[:n]


def pink_noise(sample_rate):
    """The pink-noise sound buffer of gen.voice's WIND (getPinkNoise, source/sound.cpp:7-10; draws: the declared policy) and its abs mean."""
    L = lib()
    L.ora_pink_noise.restype = C.c_int64
    am = C.c_float()
    n = L.ora_pink_noise(C.c_int32(sample_rate), None, C.c_int64(0), C.byref(am))
    out = np.zeros(n, np.float64)
    L.ora_pink_noise(C.c_int32(sample_rate), _p(out), C.c_int64(n), C.byref(am))
    return out, am.value


def wind_start_position(key, noteid, play, it, which, size):
    L = lib()
    L.ora_wind_start_position.restype = C.c_int32
    return L.ora_wind_start_position(C.c_uint64(key), C.c_int64(noteid), C.c_int32(play), C.c_int32(it), C.c_int32(which), C.c_int64(size))


class OraVoiceParams(C.Structure):
    """Mirror of oracle/oracle_voice.hpp: ora_voice_params."""
    _fields_ = [("sample_rate", C.c_int32), ("n_channels", C.c_int32), ("length_ms", C.c_float), ("length_exponent", C.c_float),
                ("xfade_length_seconds", C.c_float), ("low_freq", C.c_float), ("high_freq", C.c_float), ("sine_gain", C.c_float),
                ("loudness_level", C.c_float), ("loudness_compensation", C.c_float), ("loudness_ref_freq_index", C.c_int32),
                ("pan", C.c_float), ("random_pan", C.c_int32), ("seed", C.c_uint64),
                ("mode", C.c_int32), ("interpolation", C.c_int32), ("markov_start_node", C.c_int32), ("markov_pre_tries", C.c_int32),
                ("markov_min_path_length", C.c_int32), ("markov_additional_tries", C.c_int32), ("articulative_pause_seconds", C.c_float),
                ("freq_scatter", C.c_float), ("length_exponent_scatter", C.c_float), ("phase_ratio1", C.c_float), ("phase_ratio2", C.c_float),
                ("d1", C.c_float), ("d2", C.c_float), ("harmonic_attenuation", C.c_float), ("xfade_freq", C.c_int32),
                ("freq_transition_seconds", C.c_float), ("freq_interpolation", C.c_int32), ("seed_param", C.c_int32),
                ("pink_noise_lp_gain", C.c_float), ("pink_noise_bp_gain", C.c_float), ("pink_noise_br_gain", C.c_float),
                ("bp_octave_width_min", C.c_float), ("bp_octave_width_max", C.c_float), ("center_octave_min_long_term", C.c_float),
                ("center_octave_max_long_term", C.c_float), ("center_short_term_ratio", C.c_float), ("seconds_slow_iter_short_term", C.c_float),
                ("seconds_slow_iter_long_term", C.c_float), ("order_filters", C.c_int32)]


def voice_params_from(cfg, seed):
    """The oracle's parameters for a product ra_voice_config (same field meanings; `seed` = cfg.seed + stream)."""
    p = OraVoiceParams()
    for name, _ in OraVoiceParams._fields_:
        src = "loudness_compensation_amount" if name == "loudness_compensation" else name
        setattr(p, name, getattr(cfg, src))
    p.seed = seed
    return p


def voice_run(params, rows, n_frames):
    """gen.voice on the oracle, one stream.  rows: (sample, type, noteid, frequency, velocity).  Returns (out [n, 2] float32, dropped)."""
    rows = sorted(rows, key=lambda r: r[0])
    es = np.array([r[0] for r in rows], np.int64)
    et = np.array([r[1] for r in rows], np.int32)
    ei = np.array([r[2] for r in rows], np.int64)
    ef = np.array([r[3] for r in rows], np.float32)
    ev = np.array([r[4] for r in rows], np.float32)
    out = np.zeros((n_frames, 2), np.float32)
    L = lib()
    L.ora_voice_run.restype = C.c_int32
    d = L.ora_voice_run(C.byref(params), _p(es), _p(et), _p(ei), _p(ef), _p(ev), C.c_int64(len(rows)), C.c_int64(n_frames), _p(out))
    return out, d


def voice_trace(params, freqs, max_steps=1 << 26, spec_cap=4096, play_cap=4096):
    """Consecutive note-ons of one synth, each played out (ora_voice_trace).  Returns a list of dict(specs_d, specs_i, vol, played [m,3], done_step, gain_left)."""
    freqs = np.ascontiguousarray(freqs, np.float32)
    sd = np.zeros((spec_cap, 4), np.float64)
    si = np.zeros((spec_cap, 3), np.int32)
    sv = np.zeros(spec_cap, np.float32)
    pl = np.zeros((play_cap, 3), np.int64)
    info = np.zeros((len(freqs), 4), np.int64)
    L = lib()
    L.ora_voice_trace.restype = C.c_int32
    r = L.ora_voice_trace(C.byref(params), _p(freqs), C.c_int32(len(freqs)), C.c_int64(max_steps), _p(sd), _p(si), _p(sv), C.c_int64(spec_cap), _p(pl),
                          C.c_int64(play_cap), _p(info))
    assert r == 0, "ora_voice_trace failed (capacity / a note that does not end)"
    res, a, b = [], 0, 0
    for i in range(len(freqs)):
        ns, npl = int(info[i, 0]), int(info[i, 1])
        gain = np.array([info[i, 3]], np.int64).astype(np.int32).view(np.float32)[0]
        res.append(dict(specs_d=sd[a:a + ns].copy(), specs_i=si[a:a + ns].copy(), vol=sv[a:a + ns].copy(), played=pl[b:b + npl].copy(), done_step=int(info[i, 2]),
                        gain_left=float(gain)))
        a += ns
        b += npl
    return res


def markov_visits(which, start, normalized, seed, n_steps):
    v = np.zeros(4, np.int64)
    lib().ora_markov_visits(C.c_int32(which), C.c_int32(start), C.c_int32(1 if normalized else 0), C.c_uint64(seed), C.c_int64(n_steps), _p(v))
    return v


def itp(kind, t, b, c, d):
    t = np.ascontiguousarray(t, np.float64)
    out = np.empty_like(t)
    lib().ora_itp(C.c_int32(kind), _p(t), C.c_int64(t.size), C.c_double(b), C.c_double(c), C.c_double(d), _p(out))
    return out


def elem_run(ops, n, sample_rate=48000, ahdsr=(0, 0, 0, 0), sustain=1.0, stride=1024, with_ctrl=True):
    """One ResynthElement stepped alone (ora_elem_run).  ops: (sample, op, value), op = 0 setAngleIncrements, 1 setVolumeTarget, 2 note-on
    preparations (value = pan), 3 setStartAngle, 4 onKeyPressed(delay), 5 onKeyReleased(delay).  Returns [n, 6]: imag(0), envelope, increment,
    volume, phasor angle / pi, RT-active."""
    ops = sorted(ops, key=lambda r: r[0])
    os_ = np.array([r[0] for r in ops], np.int64)
    ot = np.array([r[1] for r in ops], np.int32)
    ov = np.array([r[2] for r in ops], np.float64)
    a4 = np.array(ahdsr, np.int32)
    out = np.zeros((n, 6), np.float64)
    lib().ora_elem_run(C.c_int32(sample_rate), _p(a4), C.c_float(sustain), C.c_int32(stride), C.c_int32(1 if with_ctrl else 0), _p(os_), _p(ot), _p(ov),
                       C.c_int64(len(ops)), C.c_int64(n), _p(out))
    return out


def logramp_run(from_, to, duration, start, interp, n):
    out = np.zeros(n, np.float64)
    lib().ora_logramp_run(C.c_double(from_), C.c_double(to), C.c_double(duration), C.c_double(start), C.c_int32(interp), C.c_int64(n), _p(out))
    return out


def limiter_run(lr):
    x = np.ascontiguousarray(lr, np.float64).copy()
    tgt = np.zeros(len(x), np.float64)
    lib().ora_limiter_run(_p(x), C.c_int64(len(x)), _p(tgt))
    return x, tgt


def det_math(x, which):
    """oracle_math.hpp: which = 'log2' | 'log10' | 'exp10' | 'pitch_to_freq' (440 * det_pow(2^(1/12), x - 69)) over a float64 array."""
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    lib().ora_det_math(x.ctypes.data, x.size, {"log2": 0, "log10": 1, "exp10": 2, "pitch_to_freq": 3}[which], out.ctypes.data)
    return out
