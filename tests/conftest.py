import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# (Round 3 ran the suite with ZKHIP_NO_PIN_WITNESS=1: the witness generator page-locked the storage of a std::vector in place.  The wire
# values now live in a mapping of their own (csrc/recursion.hip WireBuf), registered once per circuit user: the suite runs the default.)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) spends its time in the oracle's proofs of verifier circuits and in the sanitizer builds: one
    process takes ~25 minutes, four take under eight.  Where pytest-xdist is installed and the caller did not choose, `-m "not gpu"` runs
    on min(4, cores) workers (ZKHIP_PYTEST_WORKERS=n overrides, 0 = one process).  The GPU suite is never touched: one GPU, one process."""
    if getattr(config.option, "markexpr", "") != "not gpu" or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    if getattr(config.option, "numprocesses", None) or getattr(config.option, "collectonly", False) or not config.pluginmanager.hasplugin("xdist"):
        return None
    n = int(os.environ.get("ZKHIP_PYTEST_WORKERS", min(4, os.cpu_count() or 1)))
    if n > 1:
        config.option.numprocesses = n
        config.option.dist = "load"
        config.option.tx = ["popen"] * n
    return None


@pytest.fixture(scope="session")
def ora():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(autouse=True)
def _restore_context_config(request):
    """A test may change the session context's configuration (zk.set_config(jit=2)): the defaults come back afterwards."""
    yield
    if "zk" in request.fixturenames:
        import zkvm_prover_amd as z

        request.getfixturevalue("zk").set_config(z.Config.default())


@pytest.fixture(scope="session")
def zk():
    """HIP context on cuda:0; the GPU tests fail loudly if the extension is missing."""
    import torch
    import zkvm_prover_amd as z

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = z.Context(0)
    yield ctx
    ctx.close()
