import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The witness generator page-locks its (large) value array in place for the copy to the device (csrc/recursion.hip).  Inside this long-lived
# test process that is the one thing the driver is told about the process's own heap; the suite does not measure it (bench.py's guest
# flow runs the default), so the tests keep every host array pageable.
os.environ.setdefault("ZKHIP_NO_PIN_WITNESS", "1")


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
