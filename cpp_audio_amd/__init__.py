"""cpp_audio_amd — MI355X-native rt.resynth analysis -> resynthesis hot path of OlivierSohn/cpp.audio.

Only what the path needs: csrc/ (HIP kernels + C ABI, built into libresynth_amd.so), host/ (C++ shims
mirroring the reference's ComputeFunc / element interfaces), api.py (ctypes binding used by tests and bench),
signals.py (seeded synthetic inputs).  `cpp.audio_amd` at the repo root is a symlink to this directory
(a dot cannot appear in an importable package name).
"""
__all__ = ["api", "signals"]
