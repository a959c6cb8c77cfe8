import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked `gpu` ran without a GPU; select with -m 'not gpu' on CPU hosts")
    # the HIP extension must be the thing under test: build it if this checkout has no binary yet (the
    # .so is git-ignored; hipcc is on the GPU box), then fail loudly if it still cannot be loaded
    from pytextgcn_amd import _lib, build
    if not os.path.exists(build.LIB_PATH):
        build.build()
    _lib.load()
    return torch.device("cuda:0")
