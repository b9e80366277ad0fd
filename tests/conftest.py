import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ora():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def zk():
    """HIP context on cuda:0; the GPU tests fail loudly if the extension is missing."""
    import torch
    import zkvm_prover_amd as z

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = z.Context(0)
    yield ctx
    ctx.close()
