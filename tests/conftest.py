import os
import sys

import pytest

try:
    # torch bundles its own copy of the HIP runtime (libamdhip64 of its ROCm build); librts_amd.so links the system one.
    # Whichever is loaded first serves both (same soname); loaded second, torch's copy finds no device ("No HIP GPUs are
    # available").  The tests that hand torch tensors to the library therefore need torch's runtime in place before the
    # library is first loaded, whatever subset of the tests is collected.
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def rts():
    import rts_amd
    from rts_amd import api
    if not os.path.exists(rts_amd._lib.LIB_PATH):
        rts_amd.build()
    return api
