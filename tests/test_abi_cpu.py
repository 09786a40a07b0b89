"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol of include/resynth_amd.h,
its config mirror matches the C struct byte for byte, and it refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    import __graft_entry__ as g
    g.build()
    from cpp_audio_amd import api
    return api


def test_library_exports_every_declared_symbol(api):
    hdr = open(os.path.join(ROOT, "include", "resynth_amd.h")).read()
    declared = set(re.findall(r"\b(ra_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTED_SYMBOLS), declared ^ set(api.EXPORTED_SYMBOLS)
    L = api.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_struct_mirrors_match_c_layout(api, tmp_path):
    src = tmp_path / "sz.c"
    fields = [f[0] for f in api.RaConfig._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "resynth_amd.h"\nint main(){printf("%zu\\n",sizeof(ra_config));'
                   + "".join(f'printf("%zu\\n",offsetof(ra_config,{f}));' for f in fields)
                   + 'printf("%zu %zu %zu\\n",sizeof(ra_event),sizeof(ra_stats),sizeof(ra_note));'
                   + 'printf("%zu %zu %zu\\n",sizeof(ra_carrier_event),sizeof(ra_voice_config),sizeof(ra_voice_event));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    assert int(lines[0]) == C.sizeof(api.RaConfig)
    for i, f in enumerate(fields):
        assert int(lines[1 + i]) == getattr(api.RaConfig, f).offset, f
    ev, st, nt = map(int, lines[1 + len(fields)].split())
    assert ev == api.EVENT_DTYPE.itemsize and st == C.sizeof(api.RaStats) and nt == C.sizeof(api.RaNote) and nt == api.NOTE_DTYPE.itemsize
    ce, vc, ve = map(int, lines[2 + len(fields)].split())
    assert ce == C.sizeof(api.RaCarrierEvent) == api.CARRIER_EVENT_DTYPE.itemsize
    assert vc == C.sizeof(api.RaVoiceConfig) and ve == api.VOICE_EVENT_DTYPE.itemsize


def test_defaults_match_reference(api):
    c = api.default_config()   # rt.resynth.lib.cpp:943-999
    assert abs(c.min_volume - 1e-4) < 1e-10 and abs(c.nearby_distance_tones - 0.4) < 1e-7 and c.max_track_pitches == 1.0
    assert c.autotune_tolerance_pitches == 100.0 and c.stereo_spread == 1.0 and c.env_sustain_level == 1.0
    assert c.use_autotune == 0 and c.autotune_max_pitch == 150 and c.autotune_factor == 2 and c.autotune_bit_chord == 0b10010001
    assert c.vocoder_count_bands == 5 and c.vocoder_min_freq == 100.0 and c.vocoder_max_freq == 20000.0
    assert abs(c.carrier_volume - 0.1) < 1e-8 and c.analysis_volume == 0.0 and c.analysis_input_gain == 1.0
    L = api.lib()
    assert L.ra_even_window_size_frames(4096.5 / 48000, 48000) == 4096
    assert L.ra_even_window_size_frames(0.1814, 88200) == 15998
    assert L.ra_window_center_stride_frames(0.09, 88200) == 7938


def test_config_validation_errors(api):
    if api.lib().ra_device_count() > 0:
        pytest.skip("argument validation is covered on the GPU box by test_gpu_parity")
    h = C.c_void_p()
    for kw in (dict(window_size=4095), dict(window_size=0), dict(stride=0), dict(n_channels=0), dict(n_channels=1000),
               dict(window_size=65536), dict(min_volume=-1.0), dict(vocoder_count_bands=51)):
        rc = api.lib().ra_create(C.byref(api.default_config(**kw)), 1, 0, C.byref(h))
        assert rc == -1, kw
        assert api.lib().ra_last_error()


def test_no_cpu_fallback(api):
    if api.lib().ra_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(api.RaError) as e:
        api.Context(api.default_config(), 4, 0)
    assert e.value.code == -2   # RA_ENODEVICE


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "cpp_audio_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt and "oracle_lib" not in txt and "liboracle" not in txt, os.path.join(dp, f)
    hdr = open(os.path.join(ROOT, "include", "resynth_amd.h")).read()
    assert "oracle" not in hdr


def test_signals_are_seeded_and_bounded():
    from cpp_audio_amd import signals
    a = signals.partials_stream(7, 24000)
    b = signals.partials_stream(7, 24000)
    c = signals.partials_stream(8, 24000)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.dtype == np.float32 and abs(np.max(np.abs(a)) - 0.5) < 1e-6
    s = signals.speech_like_stream(3, 24000)
    assert abs(np.max(np.abs(s)) - 0.5) < 1e-6
