import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        import ldpcdecoders_jl_amd as ldpc

        return ldpc._capi.lib().ldpc_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ldpc():
    import ldpcdecoders_jl_amd as m

    return m


@pytest.fixture(scope="session")
def gpu(ldpc):
    """GPU tests must FAIL (not skip) when the HIP library or the device is missing."""
    L = ldpc._capi.lib()  # raises if libldpc_mi355x.so is not built
    assert L.ldpc_device_count() > 0, "no gfx950 device visible: -m gpu tests need the MI355X box"
    return L
