import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle_ffi

    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def bxh():
    """The HIP device layer through its C ABI; never falls back to anything."""
    from baryonyx_amd import _ffi

    _ffi.lib()
    return _ffi
