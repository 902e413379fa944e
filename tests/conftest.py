import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libcocos_hip.so; hipcc cross-compiles gfx950 without a GPU."""
    from cocosnet_amd import _lib, build
    build.build_hip(verbose=False)
    return _lib.load()
