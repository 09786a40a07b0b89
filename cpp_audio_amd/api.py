"""ctypes binding of libresynth_amd.so (the C ABI of include/resynth_amd.h).

The HIP library is the only compute path: if it is missing this module raises at import of `lib()`,
and if no GPU is visible `Context(...)` raises — there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RESYNTH_AMD_LIB: load another build of the same library (A/B tuning runs); the default is the in-tree build
LIB_PATH = os.environ.get("RESYNTH_AMD_LIB") or os.path.join(_HERE, "libresynth_amd.so")

RA_OK = 0


class RaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"resynth_amd error {code}: {msg}")
        self.code = code


class RaConfig(C.Structure):
    """Mirror of include/resynth_amd.h:ra_config (= the atomics of rt.resynth.lib.cpp:943-999)."""
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
        ("max_pitches_per_frame", C.c_int32), ("reference_order", C.c_int32),
        ("analysis_output_feedback_gain", C.c_float), ("output_delay_seconds", C.c_float),
        ("vocoder_kind", C.c_int32), ("vocoder_env_follower_cutoff_ratio", C.c_float), ("carrier_n_channels", C.c_int32),
        ("vocoder_carrier_noise_volume", C.c_float), ("vocoder_carrier_saw_volume", C.c_float),
        ("vocoder_carrier_triangle_volume", C.c_float), ("vocoder_carrier_square_volume", C.c_float),
        ("vocoder_carrier_sine_volume", C.c_float), ("vocoder_carrier_pulse_volume", C.c_float),
        ("vocoder_carrier_pulse_width", C.c_float), ("post", C.c_int32),
    ]


class RaStats(C.Structure):
    _fields_ = [("frames_analysed", C.c_int64), ("samples_rendered", C.c_int64), ("dropped_note_on", C.c_int64),
                ("error_flags", C.c_int32), ("reserved", C.c_int32), ("carrier_dropped_note_on", C.c_int64)]


class RaNote(C.Structure):
    _fields_ = [("start_frame", C.c_int32), ("end_frame", C.c_int32), ("frequency", C.c_double), ("amplitude_db", C.c_double)]


class RaCarrierEvent(C.Structure):
    """Mirror of include/resynth_amd.h:ra_carrier_event."""
    _fields_ = [("sample", C.c_int64), ("stream", C.c_int32), ("type", C.c_int32), ("key", C.c_int32), ("delay", C.c_int32),
                ("frequency", C.c_float), ("velocity", C.c_float)]


CARRIER_EVENT_DTYPE = np.dtype([("sample", "<i8"), ("stream", "<i4"), ("type", "<i4"), ("key", "<i4"), ("delay", "<i4"),
                                ("frequency", "<f4"), ("velocity", "<f4")])

EVENT_DTYPE = np.dtype([("sample", "<i8"), ("type", "<i4"), ("channel", "<i4"), ("noteid", "<i8"),
                        ("frequency", "<f4"), ("velocity", "<f4")])

# every symbol include/resynth_amd.h declares
EXPORTED_SYMBOLS = [
    "ra_last_error", "ra_device_count", "ra_default_config", "ra_even_window_size_frames", "ra_window_center_stride_frames",
    "ra_create", "ra_destroy", "ra_reset", "ra_set_hip_stream", "ra_process_device", "ra_process_device_async", "ra_join", "ra_process_host",
    "ra_synchronize",
    "ra_feed", "ra_compute", "ra_get_stats", "ra_get_events", "ra_get_frame_pitches", "ra_resynth_notes", "ra_sqmag",
    "ra_host_alloc", "ra_host_free", "ra_process_host_pipelined", "ra_set_profiling", "ra_get_kernel_ms", "ra_get_kernel_launches", "ra_get_stage_timing", "ra_set_clock_stamps", "ra_get_clock_stamps", "ra_render_note",
    "ra_carrier_events", "ra_get_carrier", "ra_set_pipeline_serial", "ra_deduce_notes", "ra_deduce_notes_batch",
    "ra_voice_default_config", "ra_voice_program", "ra_voice_expand_note", "ra_voice_pink_noise", "ra_voice_wind_start_position", "ra_voice_create", "ra_voice_destroy", "ra_voice_events", "ra_voice_process_host", "ra_voice_process_device",
    "ra_voice_process_device_async", "ra_voice_set_hip_stream", "ra_voice_synchronize", "ra_voice_get_kernel_ms", "ra_voice_dropped_note_ons",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.ra_last_error.restype = C.c_char_p
        L.ra_create.argtypes = [C.POINTER(RaConfig), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.ra_destroy.argtypes = [C.c_void_p]
        L.ra_destroy.restype = None
        L.ra_reset.argtypes = [C.c_void_p]
        L.ra_set_hip_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.ra_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_process_device_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_join.argtypes = [C.c_void_p]
        L.ra_process_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_synchronize.argtypes = [C.c_void_p]
        L.ra_feed.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
        L.ra_compute.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.ra_get_stats.argtypes = [C.c_void_p, C.POINTER(RaStats)]
        L.ra_get_events.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.ra_get_events.restype = C.c_int64
        L.ra_get_frame_pitches.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        L.ra_get_frame_pitches.restype = C.c_int64
        L.ra_resynth_notes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
        L.ra_resynth_notes.restype = C.c_int64
        L.ra_deduce_notes.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
        L.ra_deduce_notes.restype = C.c_int64
        L.ra_deduce_notes_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_deduce_notes_batch.restype = C.c_int64
        L.ra_sqmag.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
        L.ra_set_profiling.argtypes = [C.c_void_p, C.c_int32]
        L.ra_set_pipeline_serial.argtypes = [C.c_void_p, C.c_int32]
        L.ra_get_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.ra_get_kernel_launches.argtypes = [C.c_void_p, C.c_void_p]
        L.ra_get_stage_timing.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.ra_process_host_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.ra_host_alloc.argtypes = [C.c_int64]
        L.ra_host_alloc.restype = C.c_void_p
        L.ra_host_free.argtypes = [C.c_void_p]
        L.ra_host_free.restype = None
        L.ra_set_clock_stamps.argtypes = [C.c_void_p, C.c_int32]
        L.ra_get_clock_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ra_carrier_events.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ra_get_carrier.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.ra_even_window_size_frames.argtypes = [C.c_float, C.c_int]
        L.ra_window_center_stride_frames.argtypes = [C.c_float, C.c_int]
        _lib = L
    return _lib


def _check(code):
    if code < 0:
        raise RaError(code, lib().ra_last_error().decode())
    return code


def default_config(**kw):
    c = RaConfig()
    _check(lib().ra_default_config(C.byref(c)))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """n_streams independent RtResynth instances on one GPU (ra_create .. ra_destroy)."""

    def __init__(self, cfg, n_streams, device=0):
        self.cfg = cfg
        self.n_streams = int(n_streams)
        self._h = C.c_void_p()
        _check(lib().ra_create(C.byref(cfg), self.n_streams, int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().ra_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        _check(lib().ra_reset(self._h))

    def set_hip_stream(self, stream_handle):
        _check(lib().ra_set_hip_stream(self._h, C.c_void_p(stream_handle)))

    def process_host(self, voice, carrier=None, n_frames=None, out=None, chunk_frames=0):
        """voice/carrier: [n_streams, n] float32 (or None). Returns out [n_streams, n, 2] float32 (`out`: a caller's buffer, e.g. a PinnedArray's).
        chunk_frames > 0: ra_process_host_pipelined (copies and kernels of consecutive chunks overlap)."""
        a = voice if voice is not None else carrier
        if a is None:
            S, n = self.n_streams, int(n_frames)
        else:
            S, n = a.shape
        assert S == self.n_streams
        v = np.ascontiguousarray(voice, np.float32) if voice is not None else None
        c = np.ascontiguousarray(carrier, np.float32) if carrier is not None else None
        if out is None:
            out = np.empty((S, n, 2), np.float32)
        assert out.shape == (S, n, 2) and out.dtype == np.float32 and out.flags.c_contiguous
        if chunk_frames > 0:
            _check(lib().ra_process_host_pipelined(self._h, _p(v), _p(c), n, _p(out), int(chunk_frames)))
        else:
            _check(lib().ra_process_host(self._h, _p(v), _p(c), n, _p(out)))
        return out

    def process_device(self, d_voice, d_carrier, n_frames, d_out, overlap=False):
        """Raw device pointers (ints), ordered on the context's stream; call synchronize() before reading on the host.
        overlap=True (ra_process_device_async): the stream is not made to wait for the block, so the next block's analysis overlaps
        this block's rendering; the output may only be read after join() / synchronize()."""
        f = lib().ra_process_device_async if overlap else lib().ra_process_device
        _check(f(self._h, C.c_void_p(d_voice or 0), C.c_void_p(d_carrier or 0), int(n_frames), C.c_void_p(d_out)))

    def join(self):
        """Makes the context's stream wait for every block issued so far (no host synchronisation)."""
        _check(lib().ra_join(self._h))

    def synchronize(self):
        _check(lib().ra_synchronize(self._h))

    def feed(self, voice, carrier=None, stream=0):
        v = np.ascontiguousarray(voice, np.float32) if voice is not None else None
        c = np.ascontiguousarray(carrier, np.float32) if carrier is not None else None
        n = len(v) if v is not None else len(c)
        _check(lib().ra_feed(self._h, stream, _p(v), _p(c), n))

    def compute(self, buf, n_frames, stream=0):
        """Accumulates into buf ([n_frames, 2] float64), like the reference's ComputeFunc."""
        assert buf.dtype == np.float64 and buf.flags.c_contiguous
        return _check(lib().ra_compute(self._h, stream, _p(buf), n_frames))

    def carrier_events(self, rows):
        """Queue note events for the carrier synth.  rows: iterable of (sample, stream, type, key[, delay[, frequency[, velocity]]]),
        or a CARRIER_EVENT_DTYPE array; per stream in non-decreasing sample order."""
        if isinstance(rows, np.ndarray) and rows.dtype == CARRIER_EVENT_DTYPE:
            ev = np.ascontiguousarray(rows)
        else:
            ev = np.zeros(len(rows), CARRIER_EVENT_DTYPE)
            for i, r in enumerate(rows):
                r = tuple(r)
                r = r + (0, 0.0, 1.0)[len(r) - 4:] if len(r) < 7 else r
                ev[i] = r
        _check(lib().ra_carrier_events(self._h, _p(ev), len(ev)))

    def carrier(self, n_frames, stream=0):
        """The carrier the vocoder was given during the last process call (only when it came from the carrier synth)."""
        out = np.zeros(n_frames, np.float64)
        _check(lib().ra_get_carrier(self._h, stream, _p(out), n_frames))
        return out

    def set_profiling(self, enable=True):
        _check(lib().ra_set_profiling(self._h, 1 if enable else 0))

    def set_pipeline_serial(self, on=True):
        """Measurement aid: run a block's stages one after the other (each kernel alone on the chip) instead of overlapped."""
        _check(lib().ra_set_pipeline_serial(self._h, 1 if on else 0))

    STAGES = ("analysis", "track", "vocoder", "propagate", "render", "history", "prep")

    def _stage_timing(self):
        ms = np.zeros(8, np.float64)
        k = np.zeros(8, np.int64)
        calls = C.c_int64(0)
        n = _check(lib().ra_get_stage_timing(self._h, 8, _p(ms), _p(k), C.byref(calls)))
        assert n == len(self.STAGES), n
        return ms[:n].tolist(), k[:n].tolist(), calls.value

    def kernel_ms(self):
        """({'analysis','track','vocoder','propagate','render','history','prep'} -> accumulated ms, number of calls); 'prep' is the
        tracking pre-pass ra_track_prep_kernel, launched behind every analysis launch (ra_get_stage_timing)."""
        ms, _, calls = self._stage_timing()
        return dict(zip(self.STAGES, ms)), calls

    def kernel_launches(self):
        """kernel launches behind each kernel_ms() entry (a block is cut into pipeline chunks)."""
        _, k, _ = self._stage_timing()
        return dict(zip(self.STAGES, k))

    def set_clock_stamps(self, on=True):
        """Diagnostic (ra_set_clock_stamps): stamp s_memtime / s_memrealtime around every analysis / render workgroup; clears the sums."""
        _check(lib().ra_set_clock_stamps(self._h, 1 if on else 0))

    def clock_stamps(self):
        """{'analysis': (GHz, workgroups), 'render': (GHz, workgroups)}: the clock the chip held inside the two throughput kernels."""
        g = np.zeros(2, np.float64)
        k = np.zeros(2, np.int64)
        _check(lib().ra_get_clock_stamps(self._h, _p(g), _p(k)))
        return {"analysis": (float(g[0]), int(k[0])), "render": (float(g[1]), int(k[1]))}

    def stats(self):
        s = RaStats()
        _check(lib().ra_get_stats(self._h, C.byref(s)))
        return s

    def events(self, stream, cap=1 << 20):
        ev = np.zeros(cap, EVENT_DTYPE)
        n = _check(lib().ra_get_events(self._h, stream, _p(ev), cap))
        assert n <= cap
        return ev[:n]

    def frame_pitches(self, stream, frame_cap=1 << 16, item_cap=1 << 22):
        fire = np.zeros(frame_cap, np.int64)
        offs = np.zeros(frame_cap + 1, np.int64)
        pitch = np.zeros(item_cap, np.float64)
        vol = np.zeros(item_cap, np.float64)
        nf = _check(lib().ra_get_frame_pitches(self._h, stream, _p(fire), _p(offs), frame_cap, _p(pitch), _p(vol), item_cap))
        assert nf <= frame_cap and offs[nf] <= item_cap
        return fire[:nf], offs[:nf + 1], pitch[:offs[nf]], vol[:offs[nf]]


class PinnedArray:
    """A numpy array over page-locked host memory (ra_host_alloc): `with PinnedArray((S, n), np.float32) as a: a.array[...]`."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib().ra_host_alloc(self.nbytes)
        if not self._p:
            raise RaError(-3, lib().ra_last_error().decode())
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p), dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            lib().ra_host_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def sqmag(windows, window_kind=0, device=0):
    """windows [n, N] float32 -> (|X|^2 [n, L/2] float64, L)."""
    w = np.ascontiguousarray(windows, np.float32)
    n, N = w.shape
    L = 1
    while L < N:
        L <<= 1
    out = np.empty((n, L // 2), np.float64)
    fl = C.c_int32(0)
    _check(lib().ra_sqmag(_p(w), n, N, window_kind, device, _p(out), C.byref(fl)))
    assert fl.value == L
    return out, L


NOTE_DTYPE = np.dtype([("start_frame", "<i4"), ("end_frame", "<i4"), ("frequency", "<f8"), ("amplitude_db", "<f8")])


def deduce_notes(signal, sr, half_window_size, window_center_stride, window_kind=0, device=0):
    """ra_deduce_notes: mono float32 signal -> structured array of notes (start_frame, end_frame, frequency, amplitude_db)."""
    x = np.ascontiguousarray(signal, np.float32)
    n = _check(lib().ra_deduce_notes(_p(x), len(x), sr, half_window_size, window_kind, window_center_stride, device, None, 0))
    notes = np.zeros(n, NOTE_DTYPE)
    if n:
        _check(lib().ra_deduce_notes(_p(x), len(x), sr, half_window_size, window_kind, window_center_stride, device, _p(notes), n))
    return notes


def deduce_notes_batch(signals, sr, half_window_size, window_center_stride, window_kind=0, device=0):
    """ra_deduce_notes_batch: [n_signals][n_frames] float32 -> a list of note arrays, one per signal (one context, one analysis launch sequence)."""
    x = np.ascontiguousarray(signals, np.float32)
    if x.ndim != 2:
        raise ValueError("signals must be [n_signals][n_frames]")
    S, n = x.shape
    offs = np.zeros(S + 1, np.int64)
    total = _check(lib().ra_deduce_notes_batch(_p(x), S, n, sr, half_window_size, window_kind, window_center_stride, device, None, 0, _p(offs)))
    notes = np.zeros(total, NOTE_DTYPE)
    if total:
        _check(lib().ra_deduce_notes_batch(_p(x), S, n, sr, half_window_size, window_kind, window_center_stride, device, _p(notes), total, _p(offs)))
    return [notes[offs[s]:offs[s + 1]] for s in range(S)]


VOICE_BIRDS, VOICE_ROBOTS, VOICE_SWEEP, VOICE_WIND = 0, 1, 2, 3
XFADE_FREQ_NO, XFADE_FREQ_NONTRIVIAL, XFADE_FREQ_ALL = 0, 1, 2
ITP_NAMES = ["LINEAR", "PROPORTIONAL_VALUE_DERIVATIVE", "EASE_IN_QUAD", "EASE_OUT_QUAD", "EASE_INOUT_QUAD", "EASE_IN_CUBIC", "EASE_OUT_CUBIC", "EASE_INOUT_CUBIC",
             "EASE_IN_QUART", "EASE_OUT_QUART", "EASE_INOUT_QUART", "EASE_IN_QUINT", "EASE_OUT_QUINT", "EASE_INOUT_QUINT", "EASE_IN_SINE", "EASE_OUT_SINE",
             "EASE_INOUT_SINE", "EASE_IN_EXPO", "EASE_OUT_EXPO", "EASE_INOUT_EXPO", "EASE_IN_CIRC", "EASE_OUT_CIRC", "EASE_INOUT_CIRC"]
ITP = {n: i for i, n in enumerate(ITP_NAMES)}  # include/resynth_amd.h: enum ra_itp


class RaVoiceConfig(C.Structure):
    """Mirror of include/resynth_amd.h:ra_voice_config (a gen.voice program, de-normalised)."""
    _fields_ = [("sample_rate", C.c_int32), ("n_channels", C.c_int32), ("length_ms", C.c_float), ("length_exponent", C.c_float),
                ("xfade_length_seconds", C.c_float), ("low_freq", C.c_float), ("high_freq", C.c_float), ("sine_gain", C.c_float),
                ("loudness_level", C.c_float), ("loudness_compensation_amount", C.c_float), ("loudness_ref_freq_index", C.c_int32),
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


VOICE_EVENT_DTYPE = np.dtype([("sample", "<i8"), ("stream", "<i4"), ("type", "<i4"), ("noteid", "<i8"), ("frequency", "<f4"), ("velocity", "<f4")])


def _voice_cfg_set(c, kw):
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def voice_default_config(**kw):
    c = RaVoiceConfig()
    _check(lib().ra_voice_default_config(C.byref(c)))
    return _voice_cfg_set(c, kw)


def voice_program(mode, index, **kw):
    """One of the reference's own programs (gen.voice.h:700-761).  Returns (config, name)."""
    c = RaVoiceConfig()
    name = C.c_char_p()
    _check(lib().ra_voice_program(int(mode), int(index), C.byref(c), C.byref(name)))
    return _voice_cfg_set(c, kw), name.value.decode()


def voice_expand_note(cfg, frequency, stream_seed, seed_value, noteid=1, spec_cap=64, play_cap=64):
    """Host-only: the ramp specs and the schedule of one note-on (ra_voice_expand_note).  Returns dict(specs_d [n,4], specs_i [n,3], vol [n],
    played [m,4], done_step, gain_left)."""
    sd = np.zeros((spec_cap, 4), np.float64)
    si = np.zeros((spec_cap, 3), np.int32)
    sv = np.zeros(spec_cap, np.float32)
    pl = np.zeros((play_cap, 4), np.int64)
    info = np.zeros(4, np.int64)
    L = lib()
    L.ra_voice_expand_note.argtypes = [C.POINTER(RaVoiceConfig), C.c_float, C.c_uint64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int64, C.c_void_p]
    _check(L.ra_voice_expand_note(C.byref(cfg), float(frequency), int(stream_seed), int(seed_value), int(noteid), _p(sd), _p(si), _p(sv), spec_cap, _p(pl), play_cap, _p(info)))
    ns, npl = int(info[0]), int(info[1])
    gain = np.array([info[3]], np.int64).astype(np.int32).view(np.float32)[0]
    return dict(specs_d=sd[:ns], specs_i=si[:ns], vol=sv[:ns], played=pl[:npl], done_step=int(info[2]), gain_left=float(gain))


def voice_pink_noise(sample_rate):
    """Host-only: the pink-noise sound buffer of WIND and its abs mean (ra_voice_pink_noise)."""
    L = lib()
    L.ra_voice_pink_noise.restype = C.c_int64
    L.ra_voice_pink_noise.argtypes = [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    am = C.c_float()
    n = _check(L.ra_voice_pink_noise(int(sample_rate), None, 0, C.byref(am)))
    out = np.zeros(n, np.float64)
    _check(L.ra_voice_pink_noise(int(sample_rate), _p(out), n, C.byref(am)))
    return out, am.value


def voice_wind_start_position(stream_seed, noteid, play, iterator, which, buffer_length):
    L = lib()
    L.ra_voice_wind_start_position.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64]
    return _check(L.ra_voice_wind_start_position(int(stream_seed), int(noteid), int(play), int(iterator), int(which), int(buffer_length)))


class Voice:
    """n_streams gen.voice synths (SWEEP, BIRDS, ROBOTS or WIND) on one GPU (ra_voice_create .. ra_voice_destroy)."""

    def __init__(self, cfg, n_streams, device=0):
        self.n_streams = int(n_streams)
        self._h = C.c_void_p()
        L = lib()
        L.ra_voice_create.argtypes = [C.POINTER(RaVoiceConfig), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.ra_voice_destroy.argtypes = [C.c_void_p]
        L.ra_voice_destroy.restype = None
        L.ra_voice_events.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ra_voice_process_host.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_voice_process_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_voice_process_device_async.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ra_voice_set_hip_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.ra_voice_synchronize.argtypes = [C.c_void_p]
        L.ra_voice_get_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.ra_voice_dropped_note_ons.argtypes = [C.c_void_p]
        L.ra_voice_dropped_note_ons.restype = C.c_int64
        _check(L.ra_voice_create(C.byref(cfg), self.n_streams, int(device), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().ra_voice_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def events(self, rows):
        """rows: (sample, stream, type, noteid, frequency, velocity)."""
        ev = np.zeros(len(rows), VOICE_EVENT_DTYPE)
        for i, r in enumerate(rows):
            ev[i] = tuple(r)
        _check(lib().ra_voice_events(self._h, _p(ev), len(ev)))

    def process(self, n_frames):
        out = np.empty((self.n_streams, int(n_frames), 2), np.float32)
        _check(lib().ra_voice_process_host(self._h, int(n_frames), _p(out)))
        return out

    def process_device(self, n_frames, d_out_ptr):
        """d_out_ptr: device pointer to [n_streams][n_frames][2] float32 (e.g. a torch tensor's data_ptr())."""
        _check(lib().ra_voice_process_device(self._h, int(n_frames), C.c_void_p(int(d_out_ptr))))

    def process_device_async(self, n_frames, d_out_ptr):
        """Enqueues the block on the object's stream and returns (ra_voice_process_device_async); synchronize() waits."""
        _check(lib().ra_voice_process_device_async(self._h, int(n_frames), C.c_void_p(int(d_out_ptr))))

    def set_hip_stream(self, stream_ptr):
        _check(lib().ra_voice_set_hip_stream(self._h, C.c_void_p(int(stream_ptr or 0))))

    def synchronize(self):
        _check(lib().ra_voice_synchronize(self._h))

    def kernel_ms(self, reset=False):
        """({'pre': ms, 'render': ms}, launches) accumulated since the last reset (the first call switches the timing on)."""
        ms = np.zeros(2, np.float64)
        ln = np.zeros(2, np.int64)
        _check(lib().ra_voice_get_kernel_ms(self._h, _p(ms), _p(ln), 1 if reset else 0))
        return {"pre": float(ms[0]), "render": float(ms[1])}, int(ln[0])

    def dropped_note_ons(self):
        return _check(lib().ra_voice_dropped_note_ons(self._h))


def resynth_notes(start, end, freq, amp_db, stride, sr, device=0):
    notes = (RaNote * len(start))()
    for i in range(len(start)):
        notes[i] = RaNote(int(start[i]), int(end[i]), float(freq[i]), float(amp_db[i]))
    n = _check(lib().ra_resynth_notes(notes, len(start), stride, sr, device, None, 0))
    out = np.zeros(n, np.float64)
    _check(lib().ra_resynth_notes(notes, len(start), stride, sr, device, _p(out), n))
    return out
