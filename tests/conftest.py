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
    # the HIP extension must be the thing under test: fail loudly if it is absent
    from pytextgcn_amd import _lib
    _lib.load()
    return torch.device("cuda:0")
