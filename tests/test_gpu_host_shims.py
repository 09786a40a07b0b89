"""GPU test of the C++ host shims: compiles tests/host/test_shims.cpp against libresynth_amd.so, runs it, and checks the
raw outputs it leaves behind against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_shaped_drivers_run_on_the_hip_backend(oracle, tmp_path):
    lib_dir = os.path.join(ROOT, "cpp_audio_amd")
    exe = tmp_path / "test_shims"
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "test_shims.cpp"), "-o", str(exe),
                           "-L", lib_dir, "-lresynth_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    p = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    voice = np.fromfile(tmp_path / "voice.f32", np.float32)
    out_a = np.fromfile(tmp_path / "out_a.f32", np.float32).reshape(-1, 2)
    cfg = oracle.default_config(window_size=1024, stride=256, n_channels=16, analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=99)
    ref, _, _ = oracle.run_offline(cfg, voice)
    assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
    assert np.sqrt(np.mean((out_a.astype(np.float64) - ref) ** 2)) <= 1e-5
    notes = np.fromfile(tmp_path / "notes.f64", np.float64)
    nref = oracle.resynth_notes([0, 2, 2, 40, 41], [3, 2, 9, 41, 60], [440.0, 660.0, 95.0, 440.0, 3000.0], [-12.0, -20.0, -15.0, -18.0, -25.0], 400, 44100)
    assert len(notes) == len(nref) and np.sqrt(np.mean((notes - nref) ** 2)) <= 1e-9
