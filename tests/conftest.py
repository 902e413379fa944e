import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    if os.environ.get("COCOS_POISON_EMPTY") == "1":
        # every torch.empty(...) comes back filled with NaN: a kernel that reads a workspace slot it never wrote (or a caller that hands
        # over an output nobody fills) turns into a NaN in a parity test instead of depending on what the allocator's block held before
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libcocos_hip.so; hipcc cross-compiles gfx950 without a GPU."""
    from cocosnet_amd import _lib, build
    build.build_hip(verbose=False)
    return _lib.load()
