"""CPU: libzkhip.so loads and exports every symbol include/zkhip.h declares; entry points fail
loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes as C

import pytest


def test_exports_every_declared_symbol():
    import zkvm_prover_amd as z

    lib = z.load_library()
    names = z.declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libzkhip.so does not export %s" % n


def test_version_and_no_cpu_fallback():
    import torch
    import zkvm_prover_amd as z

    lib = z.load_library()
    assert lib.zkhip_version() >= 1
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU-only hosts")
    h = C.c_void_p()
    assert lib.zkhip_ctx_create(0, C.byref(h)) == -1  # ZKHIP_ERR_NO_DEVICE
    with pytest.raises(z.ZkhipError):
        z.Context(0)


def test_product_never_touches_oracle():
    """The product tree (package + include) and the measurement tools must not reference oracle/ in any way
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may)."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base in ("zkvm-prover_amd", "zkvm_prover_amd", "include", "tools"):
        for dp, _, fs in os.walk(os.path.join(root, base)):
            for f in fs:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"liboracle|oracle_lib|zk_oracle\.h|ora_[a-z]+\(", txt), os.path.join(dp, f)
