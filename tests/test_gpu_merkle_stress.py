"""GPU: the tree kernels under real concurrency, and the cause of round 4's stale node measured (docs/stale_node.md).

1. tools/ubench_barrier_race.hip (outside the prover): gfx950's s_barrier does NOT wait for a wave's LDS writes in flight -- the shape
   `ds_write ; s_barrier` lets another wave read stale LDS contents, `ds_write ; s_waitcnt lgkmcnt(0) ; s_barrier` never does.
2. tools/merkle_stress.cpp through the C ABI: tree threads rebuild their trees in place (no allocation, no synchronisation in the loop)
   beside LDS-heavy transforms on other streams; the device recomputes every plain layer from its stored children.  The kernels that ship
   must never differ; the round-4 body (zkhip_config.tree_store_early, TEST ONLY) is run the same way and what it does is reported --
   its red/green verdict is the static one of tests/test_barrier_waits_cpu.py, which does not depend on a one-in-10^7 event.
3. every form of the top of the tree (the one-workgroup top kernel from 2^3 .. 2^8 nodes, fused layers on and off, the round-4 body) gives
   one root.
4. a bounded guest-flow loop at frame 2^14 with every segment proof verified beside the proving and the device self-check on: the
   configuration that showed the stale node in 3 - 5 % of runs."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _build_stress(tmp_path_factory, lib):
    exe = tmp_path_factory.mktemp("merkle_stress_" + lib) / "merkle_stress"
    lib_dir = os.path.join(ROOT, "zkvm-prover_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "merkle_stress.cpp"), "-o", str(exe),
                           "-L", lib_dir, "-l" + lib, "-Wl,-rpath," + lib_dir, "-lpthread"])
    return str(exe)


@pytest.fixture(scope="module")
def stress_exe(tmp_path_factory):
    return _build_stress(tmp_path_factory, "zkhip")


@pytest.fixture(scope="module")
def stress_exe_test_kernels(tmp_path_factory):
    """The same program over libzkhip_test.so -- the only library that holds the round-4 bodies (csrc/Makefile, -DZKHIP_TEST_KERNELS)."""
    return _build_stress(tmp_path_factory, "zkhip_test")


def _run(exe, args, expect_clean=True, **env):
    out = subprocess.run([exe] + [str(a) for a in args], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    if expect_clean:
        assert out.returncode == 0 and rep["checks_with_a_wrong_node"] == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return rep, re.search(r"root of thread 0's tree: (.*)", out.stdout).group(1)


def test_s_barrier_does_not_wait_for_lds_writes_in_flight(tmp_path):
    exe = tmp_path / "ubench_barrier_race"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", os.path.join(ROOT, "tools", "ubench_barrier_race.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe), "4000", "1024", "16"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    racy, waited = [json.loads(l) for l in out.stdout.strip().splitlines()]
    print(racy, waited)
    assert waited["stale_reads"] == 0, waited
    assert racy["stale_reads"] > 0, "ds_write ; s_barrier gave no stale read: the premise of docs/stale_node.md does not hold on this device"


def test_layers_stay_right_under_concurrent_rebuilds(stress_exe):
    rep, _ = _run(stress_exe, [3, 4000, 12, 8, 2, 8])
    assert rep["checks"] >= 1500
    _run(stress_exe, [6, 1500, 14, 40, 2, 4])


def test_the_round4_body_under_the_same_stress_is_reported(stress_exe_test_kernels):
    rep, _ = _run(stress_exe_test_kernels, [3, 4000, 12, 8, 2, 1], expect_clean=False, ZKHIP_TREE_STORE_EARLY="1")
    print("round-4 body of the fused tree kernel under LDS-heavy neighbours:", rep)
    assert rep["early_form"] is True and rep["checks"] >= 4000


def test_every_form_of_the_tree_top_gives_one_root(stress_exe, stress_exe_test_kernels):
    roots = {_run(stress_exe, [1, 2, 13, 9, 0, 1], **{name: value})[1] for name, value in
             [("ZKHIP_TOP_MAX_LOG", "6"), ("ZKHIP_TOP_MAX_LOG", "8"), ("ZKHIP_TOP_MAX_LOG", "7"), ("ZKHIP_TOP_MAX_LOG", "3"), ("ZKHIP_COOP_MAX_LOG", "3")]}
    rep, root = _run(stress_exe_test_kernels, [1, 2, 13, 9, 0, 1], ZKHIP_TREE_STORE_EARLY="1")
    assert rep["early_form"] is True
    roots.add(root)
    assert len(roots) == 1, roots


def test_the_shipped_library_has_no_test_kernels_and_refuses_the_switch(stress_exe):
    import ctypes as C

    import zkvm_prover_amd as z

    lib = z.load_library()
    assert lib.zkhip_has_test_kernels() == 0
    ctx = z.Context(0)
    cfg = ctx.config()
    assert cfg.tree_store_early == 0
    cfg.tree_store_early = 1
    assert lib.zkhip_ctx_set_config(ctx.h, C.byref(cfg)) != 0   # ZKHIP_ERR_INVALID: the kernels are not in this library
    # ... and the variable is ignored: the run is the shipped form
    rep, _ = _run(stress_exe, [1, 2, 13, 9, 0, 1], ZKHIP_TREE_STORE_EARLY="1")
    assert rep["early_form"] is False


def test_guest_flow_loop_with_self_check(tmp_path):
    cli = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_guest_files.py"), str(tmp_path), "300000"], stdout=subprocess.DEVNULL)
    # ZKHIP_NO_RETRY: a segment proof that the self-check refuses ends the run instead of being made again (FlowOptions::retry_segments) --
    # and the flow's line must say that nothing was retried
    env = dict(os.environ, ZKHIP_VERIFY_SEGMENTS="1", ZKHIP_SELF_CHECK="1", ZKHIP_NO_RETRY="1")
    segments = 0
    for i, n in enumerate([249651, 384274, 126816, 364386, 336579, 305084]):   # (lengths of failing runs of the round-5 hunt)
        open(tmp_path / "stdin.bin", "wb").write(n.to_bytes(4, "little"))
        out = tmp_path / ("o%d" % i)
        out.mkdir()
        r = subprocess.run([cli, "prove-elf", str(tmp_path / "fib.elf"), str(tmp_path / "stdin.bin"), str(out), "-", "14"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["verified"] is True
        assert line["retry_enabled"] is False and line["segments_retried"] == 0 and line["segments_retried_detail"] == [], line
        segments += line["segments"]
    assert segments > 300


def test_a_retried_segment_is_counted_and_without_the_retry_it_ends_the_run(tmp_path):
    """FlowOptions::retry_segments: a segment proof that fails once is made again AND counted in the flow's line (segment index + message);
    with ZKHIP_NO_RETRY=1 the same failure ends the run.  The failure is injected (ZKHIP_TEST_FAIL_SEGMENT)."""
    cli = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_guest_files.py"), str(tmp_path), "300000"], stdout=subprocess.DEVNULL)
    open(tmp_path / "stdin.bin", "wb").write((60000).to_bytes(4, "little"))
    cmd = [cli, "prove-elf", str(tmp_path / "fib.elf"), str(tmp_path / "stdin.bin"), str(tmp_path), "-", "14"]
    r = subprocess.run(cmd, env=dict(os.environ, ZKHIP_TEST_FAIL_SEGMENT="3"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["verified"] is True and line["retry_enabled"] is True and line["segments_retried"] == 1, line
    assert line["segments_retried_detail"] == [{"segment": 3, "message": "injected failure (FlowOptions::fail_segment_once)"}], line
    r = subprocess.run(cmd, env=dict(os.environ, ZKHIP_TEST_FAIL_SEGMENT="3", ZKHIP_NO_RETRY="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "injected failure" in r.stderr, r.stderr[-3000:]
