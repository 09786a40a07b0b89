"""CPU checks of the host-side C++ (cpp_audio_amd/host): WAV codec against the reference's own fixture files, and that the
shim header compiles against the C ABI."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def wav_dump(tmp_path_factory):
    exe = tmp_path_factory.mktemp("host") / "wav_dump"
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "host", "wav_dump.cpp"), "-o", str(exe)])
    return str(exe)


@pytest.mark.parametrize("name,scale", [("saw_pcm_24_signed_mono.wav", 2.0 ** 31), ("saw_pcm_16_signed_stereo.wav", 2.0 ** 15), ("saw_pcm_24_signed_stereo.wav", 2.0 ** 31),
                                        ("saw_pcm_32_signed_mono.wav", 2.0 ** 31), ("saw_pcm_32_signed_stereo.wav", 2.0 ** 31), ("saw_pcm_32_float_stereo.wav", 1.0)])
def test_wav_reader_matches_reference_fixtures(wav_dump, tmp_path, name, scale):
    src = os.path.join(GOLD, name)
    raw, rt = tmp_path / "raw.bin", tmp_path / "rt.wav"
    hdr = subprocess.check_output([wav_dump, src, str(raw), str(rt)]).decode().split()
    sr, ch, bits, is_float, frames = map(int, hdr)
    sr2, data = wavfile.read(src)
    data = data.reshape(len(data), -1)
    assert (sr, ch, frames) == (sr2, data.shape[1], data.shape[0]) == (44100, ch, 65536)
    got = np.fromfile(raw, np.float64).reshape(ch, frames)
    assert np.array_equal(got, data.T.astype(np.float64) / scale)      # scipy left-justifies 24-bit into int32
    # the fixture is a 1 Hz saw ramp (the 16-bit file is dithered): at most one wrap, otherwise rising by 2/44100 per frame
    d = np.diff(got[0])
    assert np.sum(d < -0.5) <= 1
    if bits >= 24:
        assert np.all(d[d > -0.5] > 0)
    assert abs(np.median(d) - 2.0 / 44100) < 2.0 / 32768 and np.max(np.abs(d[d > -0.5] - 2.0 / 44100)) < 8.0 / 32768
    # float64 writer round trip (the format resynth.hpp writes, :29-35)
    sr3, back = wavfile.read(rt)
    assert sr3 == sr and back.dtype == np.float64 and np.array_equal(back.reshape(frames, ch).T, got)


def test_wav_reader_rejects_non_audio(wav_dump, tmp_path):
    p = subprocess.run([wav_dump, os.path.join(GOLD, "not_audio.txt"), str(tmp_path / "x")], capture_output=True)
    assert p.returncode == 1 and b"error" in p.stdout


def test_shim_header_compiles_against_the_c_abi(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "cpp_audio_amd/host/resynth_amd.hpp"\n'
                   'struct N{int startFrame,endFrame;double frequency,amplitude;};\n'
                   'int main(){ if(false){ std::vector<N> n; resynth_amd::resynth(n,400,44100,"x.wav"); resynth_amd::AmdElement e; e.step(); '
                   'resynth_amd::ComputeShim c{nullptr}; double b[2]; c(b,1);} return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, str(src)])
