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


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib
