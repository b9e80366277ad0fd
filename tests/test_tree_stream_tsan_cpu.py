"""CPU: the aggregation tree's scheduling under ThreadSanitizer (the host-only pieces under TSan).

`AggregationProver::TreeStreamT` (include/zkhip_aggregation.hpp) -- the groupers, a witness thread and a device thread per level and
device slot, the greedy fold's queue, the self-verification threads -- is a template over the prover; tests/tree_stream_tsan.cpp runs it
against a prover that makes stub proofs stating the segment range beneath them and checks, in its "witness generation", what a node
circuit would: children adjacent, in order, finished, of the kinds announced, the slot's witness buffer free, traces generated under
the slot's device lock.  Three threads push segment proofs out of order, in runs of two shapes.  Every run must end with a root over
[0, n) above the leaf nodes, the leaf nodes covering every segment once, and no report from the sanitizer; an injected witness failure
must surface as the task's error with every thread joined."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tsan") / "tree_stream_tsan")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "tree_stream_tsan.cpp"),
                    "-o", out, "-lpthread"], check=True)
    return out


def _run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    return r


@pytest.mark.parametrize("greedy", [0, 1])
@pytest.mark.parametrize("n,slots,seed", [(1, 3, 1), (2, 1, 2), (5, 2, 3), (37, 3, 4), (100, 3, 5), (64, 1, 6)])
def test_tree_stream_ends_with_a_root_over_every_segment(exe, n, slots, seed, greedy):
    r = _run(exe, n, greedy, slots, seed)
    assert r.returncode == 0 and r.stdout.startswith("ok: %d segments" % n), r.stdout + r.stderr[-2000:]


@pytest.mark.parametrize("n,greedy,slots,seed,fail_at", [(37, 1, 3, 1, 5), (37, 0, 3, 1, 2), (100, 1, 3, 3, 40), (9, 0, 2, 2, 3), (9, 1, 2, 2, 1)])
def test_a_failed_witness_is_the_tasks_error_and_every_thread_ends(exe, n, greedy, slots, seed, fail_at):
    r = _run(exe, n, greedy, slots, seed, fail_at)
    assert r.returncode == 1 and "injected failure at witness %d" % fail_at in r.stdout, r.stdout + r.stderr[-2000:]
