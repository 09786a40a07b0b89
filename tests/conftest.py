import os
import sys

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for p in (_HERE, _ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch bundles its own HIP runtime under the same SONAME as /opt/rocm's: whichever loads first serves the whole
# process.  Load torch's first so tests that use both torch tensors and libresynth_amd.so share one runtime.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


# External programs (g++, compiled host drivers, torch.distributed launches) are started by a helper process that exists before the
# first test runs: whatever order the tests run in, no process that has initialised the GPU ever forks + execs (tests/launcher.py).
_launcher = None


def pytest_sessionstart(session):
    global _launcher
    import subprocess
    _launcher = subprocess.Popen([sys.executable, "-u", os.path.join(_HERE, "launcher.py")], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    # the oracle (the checker) is built here, for the same reason: the `oracle` fixture may first be asked for after a GPU test
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def pytest_sessionfinish(session, exitstatus):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.close()
            _launcher.wait(timeout=10)
        except Exception:  # pragma: no cover
            _launcher.kill()
        _launcher = None


@pytest.fixture(scope="session")
def run_program():
    """run_program(argv, timeout=600, env=None, cwd=None) -> (returncode, stdout, stderr), executed by the session's helper process."""
    import json

    def run(argv, timeout=600, env=None, cwd=None):
        assert _launcher is not None and _launcher.poll() is None, "the launcher helper is not running"
        _launcher.stdin.write(json.dumps({"argv": [str(a) for a in argv], "timeout": timeout, "env": env, "cwd": cwd}) + "\n")
        _launcher.stdin.flush()
        res = json.loads(_launcher.stdout.readline())
        return res["rc"], res["out"], res["err"]

    return run


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
