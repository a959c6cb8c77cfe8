import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


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


@pytest.fixture(scope="session")
def c4case(cuda):
    """Config c4 (2 M nodes / 50 M edges) and the oracle's normalisation of it, shared by the tests of one session."""
    from _bigcase import BigCase
    return BigCase("c4", cuda)


@pytest.fixture(scope="session")
def c5case(cuda):
    """Config c5 (8 M nodes / 200 M edges, power law) likewise."""
    from _bigcase import BigCase
    return BigCase("c5", cuda)
