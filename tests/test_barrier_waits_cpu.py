"""No barrier of the library may be reachable with an LDS write of the wave still in flight (docs/stale_node.md).

Round 4's rare wrong Merkle node was a miscompile: on the back edge of the fused tree kernel's loop hipcc 7.2 dropped the
`s_waitcnt lgkmcnt(0)` of __syncthreads()' release fence; gfx950's s_barrier does not wait for LDS traffic by itself, so once in ~10^7
workgroups a wave read what the previous workgroup had left in the LDS.  tools/check_barriers.py re-derives the wait-count data flow from
the compiled gfx950 assembly of every source; this test runs it (no GPU: hipcc cross-compiles) and insists that
  * every kernel that ships is clean, and
  * the round-4 body kept behind zkhip_config.tree_store_early (TEST ONLY) IS flagged -- the checker sees the bug it was written for.
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CSRC = os.path.join(ROOT, "zkvm-prover_amd", "csrc")


@pytest.fixture(scope="module")
def findings():
    import check_barriers as cb

    if not os.path.exists(cb.HIPCC):
        pytest.skip("no hipcc")
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=6) as ex:   # (hipcc -S: ~5 s per file)
        return [x for found in ex.map(cb.check_file, files) for x in found]


def test_no_shipped_kernel_reaches_a_barrier_with_an_lds_write_in_flight(findings):
    errors = [f for f in findings if f[3] == "w" and "_early" not in f[1]]
    assert not errors, "\n".join("%s: %s block %s after `%s`" % (f[0], f[1], f[2], f[4]) for f in errors)


def test_the_round4_body_of_the_fused_tree_kernel_is_flagged(findings):
    early = [f for f in findings if f[3] == "w" and "k_compress_coop_multi_early" in f[1]]
    assert early, "the checker no longer sees the lost wait of the round-4 kernel (compiler changed? then the A/B kernel has lost its point)"


def test_the_data_flow_on_hand_written_shapes():
    import check_barriers as cb

    loop_without_wait = ["s_waitcnt lgkmcnt(0)", ".LBB0_1:", "s_barrier", "ds_read_b32 v1, v1", "s_waitcnt lgkmcnt(0)", "ds_write_b32 v0, v1",
                         "s_cbranch_scc0 .LBB0_1", "s_endpgm"]
    assert [k for _, k, _ in cb.check_kernel(loop_without_wait)] == ["w"]
    loop_with_wait = [ln for ln in loop_without_wait]
    loop_with_wait.insert(6, "s_waitcnt lgkmcnt(0)")
    assert cb.check_kernel(loop_with_wait) == []
    read_only = ["ds_read_b32 v1, v1", "s_barrier", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert [k for _, k, _ in cb.check_kernel(read_only)] == ["r"]
    lane_moves = ["ds_bpermute_b32 v1, v2, v3", "s_barrier", "s_endpgm"]   # no LDS memory touched
    assert cb.check_kernel(lane_moves) == []
    partial_wait = ["ds_write_b32 v0, v1", "ds_write_b32 v2, v3", "s_waitcnt lgkmcnt(1)", "s_barrier", "s_endpgm"]
    assert [k for _, k, _ in cb.check_kernel(partial_wait)] == ["w"]
