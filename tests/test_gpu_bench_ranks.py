"""GPU (one device): `ZKHIP_BENCH_DRYRUN_1GPU=1 python bench.py --gpus 2` -- bench.py starts two ranks itself, both prove
real proofs on cuda:0, exchange them asynchronously (gloo on host tensors, since one GPU cannot host two RCCL ranks),
rank 0 verifies and prints ONE JSON line with n_gpus = 2.  A plumbing check of the N-rank path, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, ZKHIP_BENCH_DRYRUN_1GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1",
                        "--log-rows", "16", "--width", "64", "--inflight", "2"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["verified"] is True
    assert out["config"]["exchange"]["posted_per_rank"] == 6
    assert "DRY RUN" in out["config"]["parallelism"]
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only


def test_bench_under_torch_distributed_run_on_one_gpu():
    """The driver's own launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`), with both ranks on cuda:0 (dry run): bench.py must use the ranks it is given."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, ZKHIP_BENCH_DRYRUN_1GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--log-rows", "14", "--width", "40", "--inflight", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["verified"] is True and out["config"]["exchange"]["posted_per_rank"] == 4
