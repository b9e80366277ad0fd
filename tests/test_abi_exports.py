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


def test_the_racy_ab_kernels_are_in_the_test_library_only():
    """The round-4 bodies of the fused tree kernels (docs/stale_node.md) are compiled under -DZKHIP_TEST_KERNELS into libzkhip_test.so; the
    library that ships holds neither the kernels nor a way to select them."""
    import os

    import zkvm_prover_amd as z

    lib = z.load_library()
    assert lib.zkhip_has_test_kernels() == 0
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zkvm-prover_amd")
    shipped = open(os.path.join(here, "libzkhip.so"), "rb").read()
    assert b"k_compress_top_early" not in shipped and b"k_compress_coop_multi_early" not in shipped
    test_path = os.path.join(here, "libzkhip_test.so")
    test = open(test_path, "rb").read()
    assert b"k_compress_top_early" in test and b"k_compress_coop_multi_early" in test
    tlib = C.CDLL(test_path)
    assert tlib.zkhip_has_test_kernels() == 1
    for n in z.declared_symbols():
        assert hasattr(tlib, n), "libzkhip_test.so does not export %s" % n


def test_environment_overrides_are_clamped_into_the_accepted_ranges():
    """zkhip_config_default with out-of-range ZKHIP_* values gives a configuration zkhip_ctx_set_config's range check would accept
    (ADVICE round 5); ZKHIP_TREE_STORE_EARLY is ignored by the shipped library."""
    import os
    import subprocess
    import sys

    code = ("import zkvm_prover_amd as z, ctypes as C\n"
            "from zkvm_prover_amd._binding import Config\n"
            "c = Config(); z.load_library().zkhip_config_default(C.byref(c))\n"
            "print(c.hash_block, c.ntt_log_lanes, c.quot_streams, c.top_max_log, c.coop_max_log, c.tree_store_early, c.jit_min_log_work, c.commit_parts)\n")
    env = dict(os.environ, ZKHIP_HASH_BLOCK="1000", ZKHIP_NTT_LOG_LANES="3", ZKHIP_QUOT_STREAMS="99", ZKHIP_TOP_MAX_LOG="12", ZKHIP_COOP_MAX_LOG="40",
               ZKHIP_TREE_STORE_EARLY="1", ZKHIP_JIT_MIN_LOG_WORK="-5", ZKHIP_COMMIT_PARTS="77")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["768", "8", "4", "8", "27", "0", "0", "8"], out.stdout
