"""CPU: FoldLine (include/zkhip_prover.hpp), the bookkeeping of the aggregation tree without a fixed shape (one aggregation key: any
ADJACENT node proofs fold).  tests/fold_line_cpp.cpp drives it through simulated streams; here the properties every run must have:
children of a fold are adjacent, in order and finished before it starts; every fold but the last is full; the root is an internal
node over all leaf nodes; the number of internal nodes is the balanced tree's; and the shape follows the timing -- a comb when the
folds keep up (ONE fold after the last leaf node), the balanced tree's depth when they do not."""
import math
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fold") / "fold_line_cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "fold_line_cpp.cpp"), "-o", exe], check=True)

    def run(m, arity, leaf_gap, fold_time, workers, seed=0):
        out = subprocess.run([exe, str(m), str(arity), str(leaf_gap), str(fold_time), str(workers), str(seed)], check=True,
                             capture_output=True, text=True).stdout.splitlines()
        folds = []
        for ln in out[:-1]:
            w = ln.split()
            assert w[0] == "fold"
            n = int(w[6])
            kinds = [int(x) for x in w[7:7 + n]]
            rng = [int(x) for x in w[7 + n:7 + 3 * n]]
            folds.append(dict(start=float(w[1]), end=float(w[2]), lo=int(w[3]), hi=int(w[4]), depth=int(w[5]), kinds=kinds,
                              kids=[(rng[2 * i], rng[2 * i + 1]) for i in range(n)]))
        w = out[-1].split()
        assert w[0] == "root", out[-1]
        root = dict(lo=int(w[1]), hi=int(w[2]), depth=int(w[3]), end=float(w[4]), t_last=float(w[5]), folds=int(w[6]))
        return folds, root
    return run


def check(m, arity, folds, root, leaf_gap, seed):
    assert (root["lo"], root["hi"]) == (0, m)
    assert root["folds"] == len(folds) == max(1, math.ceil((m - 1) / (arity - 1)))
    made = {}   # range -> (time it exists, depth)
    for f in folds:
        assert 1 <= len(f["kids"]) <= arity
        assert f["kids"][0][0] == f["lo"] and f["kids"][-1][1] == f["hi"]
        for (a, b), (c, d) in zip(f["kids"], f["kids"][1:]):
            assert b == c                                   # adjacent, in order
        depth = 0
        for (a, b), kind in zip(f["kids"], f["kinds"]):
            if kind:                                        # a leaf node: one leaf, its shape's kind
                assert b == a + 1 and kind == 1 + a % 2
                if not seed:
                    assert leaf_gap * (a + 1) <= f["start"] + 1e-9
            else:                                           # an earlier fold's proof, finished
                t, dep = made[(a, b)]
                assert t <= f["start"] + 1e-9
                depth = max(depth, dep)
        assert f["depth"] == depth + 1
        made[(f["lo"], f["hi"])] = (f["end"], f["depth"])
    for f in folds[:-1]:
        assert len(f["kids"]) == arity                      # only the last fold may be short
    assert made[(0, m)][1] == root["depth"]
    # every leaf node is under the root exactly once
    covered = sorted(k for f in folds for k, kind in zip(f["kids"], f["kinds"]) if kind)
    assert covered == [(k, k + 1) for k in range(m)]


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 9, 15, 16, 40, 101])
@pytest.mark.parametrize("arity", [2, 3, 4])
@pytest.mark.parametrize("fold_time,workers,seed", [(65.0, 2, 0), (400.0, 2, 0), (65.0, 1, 11), (150.0, 3, 5)])
def test_every_stream_folds_to_one_internal_root(sim, m, arity, fold_time, workers, seed):
    folds, root = sim(m, arity, 40.5, fold_time, workers, seed)
    check(m, arity, folds, root, 40.5, seed)


def test_folds_that_keep_up_leave_one_fold_after_the_last_leaf_node(sim):
    # the guest flow's case: a leaf node every 40 ms, a fold 65 ms: root = (everything before, the last two leaf nodes)
    folds, root = sim(15, 3, 40.5, 65.0, 2)
    assert root["end"] - root["t_last"] == pytest.approx(65.0)
    assert folds[-1]["kinds"] == [0, 2, 1] and folds[-1]["kids"] == [(0, 13), (13, 14), (14, 15)]
    # the fixed grouping (15 -> 5 -> 2 -> 1) has three folds in a row after the last leaf node


def test_folds_that_lag_give_the_balanced_depth(sim):
    folds, root = sim(27, 3, 10.0, 500.0, 2)
    assert root["depth"] == 3          # 27 -> 9 -> 3 -> 1
    folds, root = sim(81, 3, 1.0, 500.0, 4)
    assert root["depth"] <= 5          # (the balanced tree's 4, plus what arrival order costs)
