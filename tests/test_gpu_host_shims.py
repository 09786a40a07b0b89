"""GPU test of the C++ host shims: compiles tests/host/test_shims.cpp against libresynth_amd.so, runs it, and checks the
raw outputs it leaves behind against the CPU oracle.  The compiler and the compiled programs are started by the session's helper
process (conftest.run_program), never by this — possibly GPU-initialised — process."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_shaped_drivers_run_on_the_hip_backend(oracle, tmp_path, run_program):
    lib_dir = os.path.join(ROOT, "cpp_audio_amd")
    exe = tmp_path / "test_shims"
    rc, so, se = run_program(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "test_shims.cpp"), "-o", str(exe),
                              "-L", lib_dir, "-lresynth_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert rc == 0, so + se
    rc, so, se = run_program([str(exe), str(tmp_path)])
    assert rc == 0, so + se
    voice = np.fromfile(tmp_path / "voice.f32", np.float32)
    out_a = np.fromfile(tmp_path / "out_a.f32", np.float32).reshape(-1, 2)
    cfg = oracle.default_config(window_size=1024, stride=256, n_channels=16, analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0, seed=99)
    ref, _, _ = oracle.run_offline(cfg, voice)
    assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
    assert np.sqrt(np.mean((out_a.astype(np.float64) - ref) ** 2)) <= 1e-5
    notes = np.fromfile(tmp_path / "notes.f64", np.float64)
    nref = oracle.resynth_notes([0, 2, 2, 40, 41], [3, 2, 9, 41, 60], [440.0, 660.0, 95.0, 440.0, 3000.0], [-12.0, -20.0, -15.0, -18.0, -25.0], 400, 44100)
    assert len(notes) == len(nref) and np.sqrt(np.mean((notes - nref) ** 2)) <= 1e-9


def test_offline_job_file_runs_end_to_end(oracle, tmp_path, run_program):
    """rt.resynth.job flow (params.cpp:183-386, lib.cpp:1185-1235): job config JSON -> WAV in -> preset JSON -> GPU -> stereo float32 WAV."""
    import json
    from scipy.io import wavfile
    from cpp_audio_amd import signals
    lib_dir = os.path.join(ROOT, "cpp_audio_amd")
    exe = tmp_path / "preset_tool"
    rc, so, se = run_program(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "preset_tool.cpp"), "-o", str(exe),
                              "-L", lib_dir, "-lresynth_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert rc == 0, so + se
    sr = 48000
    voice = np.ascontiguousarray(signals.make_streams(1, int(1.5 * sr), sr, base_seed=5, kind="speech")[0], np.float32)
    wavfile.write(tmp_path / "voice.wav", sr, voice)
    job = {"preset_file": os.path.join(ROOT, "tests", "golden", "preset_example.json"), "input_voice_file": str(tmp_path / "voice.wav"),
           "input_carrier_file": "", "output_file": str(tmp_path / "out.wav"), "post": "none"}
    (tmp_path / "job.json").write_text(json.dumps(job))
    rc, so, se = run_program([str(exe), "job", str(tmp_path / "job.json"), "16", "77"])
    assert rc == 0, so + se
    st = dict(l.split() for l in so.strip().splitlines())
    assert int(st["error_flags"]) == 0 and int(st["samples_rendered"]) == len(voice)
    sr2, out = wavfile.read(tmp_path / "out.wav")
    assert sr2 == sr and out.dtype == np.float32 and out.shape == (len(voice), 2)
    # the same parameters through the CPU restatement (the preset's values, restated)
    cfg = oracle.default_config(sample_rate=sr, window_size=1022, stride=256, n_channels=16, analysis_volume=1.0, stereo_spread=0.0,
                                carrier_volume=0.0, seed=77, use_autotune=1, autotune_scale_mode=2, autotune_root_note=7,
                                pitch_shift_pre_autotune=-12.0, env_attack_seconds=0.0020000000949949026,
                                env_release_seconds=0.009999999776482582, autotune_bit_chord=18446744073709551557)
    ref, _, _ = oracle.run_offline(cfg, voice)
    assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-4
    assert np.sqrt(np.mean((out.astype(np.float64) - ref) ** 2)) <= 1e-5


def test_one_host_thread_and_one_hip_stream_per_device_in_one_process(tmp_path, run_program):
    """SURVEY.md section 8(e)'s wording: the C++ example tests/host/device_threads.cpp drives one context per host thread, each on a HIP stream of
    its own (device t mod the device count: both on device 0 of this one-GPU box), on contiguous shards with seeds = seed + global stream index;
    every thread's output equals the same shard run from a single thread bit for bit (2 and 3 threads), and the union equals one context over
    all streams up to the rounding of the block cut."""
    from cpp_audio_amd import api, signals
    lib_dir = os.path.join(ROOT, "cpp_audio_amd")
    exe = tmp_path / "device_threads"
    rc, so, se = run_program(["g++", "-std=c++17", "-O1", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "host", "device_threads.cpp"),
                              "-o", str(exe), "-L", lib_dir, "-lresynth_amd", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert rc == 0, so + se
    S, n = 7, 40000
    x = signals.make_streams(S, n, base_seed=41)
    x.tofile(tmp_path / "voice.f32")
    kw = dict(window_size=1024, stride=256, n_channels=32, analysis_volume=1.0, stereo_spread=0.0, carrier_volume=0.0)
    for threads in (2, 3):
        rc, so, se = run_program([str(exe), str(tmp_path), S, n, threads])
        assert rc == 0, so + se
        per = (S + threads - 1) // threads
        for t in range(threads):
            lo, hi = min(S, t * per), min(S, (t + 1) * per)
            if hi <= lo:
                continue
            got = np.fromfile(tmp_path / f"out_{t}.f32", np.float32).reshape(hi - lo, n, 2)
            # the same shard, the same two ragged blocks, from this (single) thread: what a thread of the example computes does not depend on its neighbours
            cut = n // 3 + 7 * t
            with api.Context(api.default_config(seed=1234 + lo, **kw), hi - lo, 0) as ctx:
                ref = np.concatenate([ctx.process_host(x[lo:hi, :cut]), ctx.process_host(x[lo:hi, cut:])], axis=1)
            assert np.sqrt(np.mean(ref.astype(np.float64) ** 2)) > 1e-3
            assert np.array_equal(got, ref), f"{threads} threads, thread {t}"
            # ... and equals the streams' results in one context over all of them, up to the block cut's rounding
            if t == 0:
                with api.Context(api.default_config(seed=1234, **kw), S, 0) as ctx:
                    whole = ctx.process_host(x)
            assert np.max(np.abs(got.astype(np.float64) - whole[lo:hi])) < 1e-6
