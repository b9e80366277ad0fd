"""CPU: `python bench.py --gpus 2` starts its own two ranks (no torch.distributed.run), they rendezvous on 127.0.0.1,
run the asynchronous exchange of zkvm-prover_amd/shard.py end to end and rank 0 prints ONE JSON line with n_gpus = 2.
ZKHIP_BENCH_PLUMBING_ONLY=1 replaces the device pipelines with stub byte strings (there is no GPU here): this checks the
launcher and the N>1 data path, not the prover.  The same path with real proofs on one GPU:
`ZKHIP_BENCH_DRYRUN_1GPU=1 python bench.py --gpus 2` (tests/test_gpu_bench_ranks.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, ZKHIP_BENCH_PLUMBING_ONLY="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=300)
    return r


@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_own_ranks(n):
    r = _run({}, "--gpus", str(n), "--steps", "5", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 5 and out["config"]["exchange_ok"] is True
    # gather -> aggregate on rank 0: the gathered (stub) proofs of the last step folded through the one-key tree shape
    assert out["aggregate"]["proofs_folded"] == n and out["aggregate"]["nodes_per_level"][0] == (n + 3) // 4 and out["aggregate"]["nodes_per_level"][-1] == 1
    assert out["scaling"] == "weak" and out["value"] is None  # plumbing only: no measurement is claimed
    # SURVEY.md 8(e)(ii) beside (i): ONE guest task over the N devices, run by the launcher before its ranks exist and handed to rank 0
    gd = out["guest_flow_devices"]
    assert gd["plumbing_only"] is True and gd["devices"] == list(range(n)) and ("ZKHIP_DEVICES=" + ",".join(map(str, range(n)))) in gd["command"]


def test_bench_under_a_launcher_uses_the_given_ranks():
    """torch.distributed.run-style environment: bench.py must NOT spawn again."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, ZKHIP_BENCH_PLUMBING_ONLY="1", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2
    # under a launcher rank 0 runs the one-task-over-N-devices flow before it loads torch; rank 1 waited for its flag file
    assert line["guest_flow_devices"]["devices"] == [0, 1] and os.path.exists("/tmp/zkhip_bench_devices_%d.json" % port)
    os.remove("/tmp/zkhip_bench_devices_%d.json" % port)
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]


def _run_real(extra_env, *argv):
    env = dict(os.environ, **extra_env)
    for k in ("WORLD_SIZE", "RANK", "ZKHIP_BENCH_PLUMBING_ONLY", "ZKHIP_BENCH_DRYRUN_1GPU"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=600)


def test_launcher_counts_devices_without_touching_hip():
    """The NON-dry launcher branch: devices are counted from the KFD topology (stubbed here), torch is never imported in
    the parent (bench.py asserts it), and too few devices is refused before any rank starts."""
    r = _run_real({"ZKHIP_BENCH_DEVICE_COUNT": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 2 and "1 GPU(s) visible" in r.stderr, (r.returncode, r.stderr[-500:])


def test_launcher_real_branch_starts_fresh_ranks_that_fail_loudly_without_a_gpu():
    """With enough (stubbed) devices the launcher starts its ranks as fresh processes; on this GPU-less box each rank
    fails on its own (no CPU fallback) and the launcher returns that failure instead of hanging."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: this is the no-GPU failure path")
    r = _run_real({"ZKHIP_BENCH_DEVICE_COUNT": "2"}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode not in (0, 2), (r.returncode, r.stderr[-800:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_visible_gpu_count_reads_sysfs_only(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    monkeypatch.setenv("ZKHIP_BENCH_DEVICE_COUNT", "5")
    assert bench.visible_gpu_count() == 5
    monkeypatch.delenv("ZKHIP_BENCH_DEVICE_COUNT")
    assert bench.visible_gpu_count() >= 0   # whatever this box has; must not raise and must not need torch
