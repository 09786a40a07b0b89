"""CPU checks of cpp_audio_amd/host/preset.hpp: the reference's preset / job-config JSON containers
(source/rt.resynth.lib.params.cpp:5-247) and RtResynth::restorePreset / saveAsPreset (source/rt.resynth.lib.cpp:1941-2145).

The reference repository ships no preset or job file and no test of these containers, so tests/golden/preset_example.json was
written by hand in the layout nlohmann::json gives Preset::tojson (five typed maps, keys sorted, floats printed as the double
nearest the float).  Parity of the key set is checked against the list in lib.cpp:1941-2041 restated below."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
LIB_DIR = os.path.join(ROOT, "cpp_audio_amd")

# every key saveAsPreset writes (lib.cpp:1941-2041), by the typed map it lands in (member types, lib.cpp:943-999)
REF_KEYS = {
    "bool_params": ["use_autotune"],
    "enum_params": ["autotune_type", "autotune_musical_scale_mode", "autotune_musical_scale_root_note", "autotune_chord_frequencies"],
    "int32_params": ["autotune_max_pitch", "autotune_factor", "autotune_root_note_halftones_transpose", "vocoder_count_bands"],
    "uint64_params": ["autotune_bit_chord"],
    "float_params": ["pitch_wheel_multiplier", "window_size_seconds", "window_center_stride_seconds", "min_volume", "nearby_distance_tones",
                     "max_track_pitches", "autotune_tolerance_pitches", "pitch_shift_pre_autotune", "pitch_shift_post_autotune",
                     "pitch_harmonize_pre_autotune", "pitch_harmonize_post_autotune", "stereo_spread", "env_attack_seconds", "env_hold_seconds",
                     "env_decay_seconds", "env_release_seconds", "env_sustain_level", "vocoder_carrier_noise_volume", "vocoder_carrier_saw_volume",
                     "vocoder_carrier_triangle_volume", "vocoder_carrier_square_volume", "vocoder_carrier_sine_volume", "vocoder_carrier_pulse_volume",
                     "vocoder_carrier_pulse_width", "vocoder_env_follower_cutoff_ratio", "vocoder_modulator_window_size_seconds",
                     "vocoder_stride_seconds", "vocoder_min_freq", "vocoder_max_freq", "voice_volume", "carrier_volume", "vocoder_volume",
                     "analysis_volume", "analysis_input_gain", "analysis_output_feedback_gain", "output_delay_seconds"],
}


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    if not os.path.exists(os.path.join(LIB_DIR, "libresynth_amd.so")):
        pytest.fail("libresynth_amd.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = tmp_path_factory.mktemp("preset") / "preset_tool"
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "preset_tool.cpp"), "-o", str(exe),
                           "-L", LIB_DIR, "-lresynth_amd", f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib"])
    return str(exe)


def kv(text):
    return dict(line.split(" ", 1) for line in text.strip().splitlines())


def test_restore_preset_maps_every_key(tool):
    out = kv(subprocess.check_output([tool, "parse", os.path.join(GOLD, "preset_example.json"), "48000"], stderr=subprocess.DEVNULL).decode())
    # seconds -> frames: getEvenWindowSizeFrames (lib.cpp:484-487) and the stride lambda (lib.cpp:1577-1578)
    w, s = 0.02133333310484886, 0.005333333276212215
    assert int(out["window_size"]) == 2 * max(1, int(0.5 * w * 48000)) == 1022
    assert int(out["stride"]) == max(1, int(0.5 + s * 48000)) == 256
    assert out["use_autotune"] == "1" and out["autotune_scale_mode"] == "2" and out["autotune_root_note"] == "7"
    assert out["autotune_bit_chord"] == "18446744073709551557"          # a uint64 beyond 2^53 survives the reader
    assert float(out["pitch_shift_pre_autotune"]) == -12.0 and float(out["stereo_spread"]) == 0.0
    assert float(out["env_release_seconds"]) == pytest.approx(0.01, rel=1e-6) and float(out["analysis_volume"]) == 1.0
    assert int(out["n_channels"]) == 64                                   # not a preset key: untouched


def test_missing_key_keeps_the_current_value_and_is_reported(tool, tmp_path):
    doc = json.load(open(os.path.join(GOLD, "preset_example.json")))
    del doc["float_params"]["min_volume"]
    p = tmp_path / "p.json"
    p.write_text(json.dumps(doc))
    r = subprocess.run([tool, "parse", str(p), "48000"], capture_output=True, text=True)
    assert r.returncode == 0 and "Preset param 'min_volume' not found" in r.stderr     # Params::read, params.cpp:18-27
    assert float(kv(r.stdout)["min_volume"]) == pytest.approx(1e-4, rel=1e-6)              # ra_default_config's value


def test_missing_section_is_an_error_and_feedback_keys_map(tool, tmp_path):
    doc = json.load(open(os.path.join(GOLD, "preset_example.json")))
    bad = dict(doc)
    del bad["uint64_params"]                                              # Preset::fromjson uses json::at -> throws
    p = tmp_path / "bad.json"
    p.write_text(json.dumps(bad))
    r = subprocess.run([tool, "parse", str(p), "48000"], capture_output=True, text=True)
    assert r.returncode == 1 and "uint64_params" in r.stdout
    doc["float_params"]["analysis_output_feedback_gain"] = 0.5
    doc["float_params"]["output_delay_seconds"] = 0.25
    p.write_text(json.dumps(doc))
    out = kv(subprocess.check_output([tool, "parse", str(p), "48000"], stderr=subprocess.DEVNULL).decode())
    assert float(out["analysis_output_feedback_gain"]) == 0.5 and float(out["output_delay_seconds"]) == 0.25


def test_preset_round_trip_and_key_set(tool, tmp_path):
    out = tmp_path / "rt.json"
    subprocess.check_call([tool, "roundtrip", os.path.join(GOLD, "preset_example.json"), str(out)])
    a, b = json.load(open(os.path.join(GOLD, "preset_example.json"))), json.load(open(out))
    assert a.keys() == b.keys()
    for sec in a:
        assert a[sec].keys() == b[sec].keys()
        for k in a[sec]:
            assert b[sec][k] == pytest.approx(a[sec][k], rel=1e-7), (sec, k)
    assert b["uint64_params"]["autotune_bit_chord"] == 18446744073709551557
    # saveAsPreset writes exactly the reference's key set
    saved = tmp_path / "saved.json"
    subprocess.check_call([tool, "save", "48000", str(saved)])
    s = json.load(open(saved))
    for sec, keys in REF_KEYS.items():
        assert sorted(s[sec].keys()) == sorted(keys), sec
    assert s["float_params"]["window_size_seconds"] == pytest.approx(4096 / 48000) and s["float_params"]["vocoder_carrier_square_volume"] == 1.0


def test_job_config(tool, tmp_path):
    j = tmp_path / "job.json"
    j.write_text(json.dumps({"preset_file": "p.json", "input_voice_file": "v.wav", "input_carrier_file": "", "output_file": "o.wav", "post": "none"}))
    out = subprocess.check_output([tool, "jobcfg", str(j)]).decode()
    head = kv("\n".join(l for l in out.splitlines()[:5]))
    assert head["preset_file"] == "p.json" and head["post"] == "none" and head["output_file"] == "o.wav"
    assert json.loads("\n".join(out.splitlines()[5:])) == json.load(open(j))          # tojson round trip
    j.write_text(json.dumps({"preset_file": "p.json", "input_voice_file": "v.wav", "input_carrier_file": "", "output_file": "o.wav", "post": "loud"}))
    r = subprocess.run([tool, "jobcfg", str(j)], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown post" in r.stdout
    j.write_text(json.dumps({"preset_file": "p.json", "input_voice_file": "v.wav", "output_file": "o.wav", "post": "none"}))
    r = subprocess.run([tool, "jobcfg", str(j)], capture_output=True, text=True)
    assert r.returncode == 1 and "input_carrier_file" in r.stdout
