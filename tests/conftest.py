import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding, on a box with a usable GPU.  Fails (does not skip) if the library is missing."""
    from ocean_model_grid_generator_amd import _lib
    lib = _lib.load()
    assert _lib.device_count() >= 1, "no HIP device visible"
    return _lib
