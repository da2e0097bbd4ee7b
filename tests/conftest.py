import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Make sure libdgp_hip.so exists (hipcc cross-compiles without a GPU)."""
    from deepgraphpose_amd import build
    return build.build(verbose=False)


@pytest.fixture(scope="session")
def tuning_build(lib_built):
    """Tests of the opt-in paths that measured slower (256-row tile, the trainer's fast pass): they exist only in a library built with
    DGP_BUILD_FLAGS=-DDGP_TUNING; the product build skips them."""
    from deepgraphpose_amd import _lib
    if not _lib.load().dgp_tuning_build():
        pytest.skip("opt-in path of -DDGP_TUNING builds (DGP_BUILD_FLAGS=-DDGP_TUNING python -m deepgraphpose_amd.build)")
    return True
