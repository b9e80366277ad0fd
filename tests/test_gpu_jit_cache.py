"""GPU: the on-disk cache of compiled constraint kernels (ZKHIP_JIT_CACHE_DIR).  A second process with the same AIR set
loads the stored code objects and produces the same proof bytes; a corrupted cache file is ignored, not loaded."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, sys, time
import numpy as np
import zkvm_prover_amd as z
from zkvm_prover_amd import air
ctx = z.Context(0)
sa = air.SyntheticAir(width=12, n_free=5, n_bool=2, n_boundary=1, seed=4)
tr, pv = sa.gen_trace(8, seed=1)
ftr, fpv = air.fibonacci_trace(6)
airs = [dict(program=sa.program(), log_height=8, width=12, n_pvs=len(pv), trace=tr, pvs=pv),
        dict(program=air.fibonacci_air().program(), log_height=6, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
params = (1, 0, 6, 2, 3)
t0 = time.time()
pk = z.ProvingKey(ctx, params, airs)
dt = time.time() - t0
proof = pk.prove([ctx.upload(a["trace"].reshape(-1)) for a in airs], [a["pvs"] for a in airs])
assert z.verify(params, airs, [a["pvs"] for a in airs], proof) == 0
print("RESULT", hashlib.sha256(proof).hexdigest(), "%.3f" % dt)
"""


def _run(cache_dir):
    env = dict(os.environ, ZKHIP_JIT_CACHE_DIR=str(cache_dir), ZKHIP_FORCE_JIT="1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    return line[1], float(line[2])


def test_disk_cache_round_trip(tmp_path):
    d1, t_cold = _run(tmp_path)
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".hsaco"))
    assert len(files) == 2, files  # one kernel per AIR
    d2, t_warm = _run(tmp_path)
    assert d1 == d2
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".hsaco")) == files
    # a damaged file (same name, wrong content) is recompiled around, and replaced
    victim = os.path.join(tmp_path, files[0])
    with open(victim, "r+b") as f:
        f.seek(9)
        f.write(b"\x00garbage\x00")
    d3, _ = _run(tmp_path)
    assert d3 == d1
    print("keygen cold %.2f s, warm %.2f s" % (t_cold, t_warm))


TILED_SCRIPT = r"""
import hashlib
import numpy as np
import zkvm_prover_amd as z
from zkvm_prover_amd import air
ctx = z.Context(0)
sa = air.SyntheticAir(width=90, n_free=24, n_bool=4, n_boundary=3, seed=7)
tr, pv = sa.gen_trace(11, seed=2)
airs = [dict(program=sa.program(), log_height=11, width=90, n_pvs=len(pv), trace=tr, pvs=pv)]
params = (1, 0, 6, 2, 3)
pk = z.ProvingKey(ctx, params, airs)
proof = pk.prove([ctx.upload(tr.reshape(-1))], [pv])
assert z.verify(params, airs, [pv], proof) == 0
print("RESULT", hashlib.sha256(proof).hexdigest())
"""


def test_lds_tiled_constraint_kernel_gives_the_same_proof():
    """The opt-in LDS-tiled form of the compiled constraint kernel (ZKHIP_JIT_TILE=1: a workgroup copies its 64 rows into LDS once and
    eight waves share the constraints) against the plain class form and the interpreter: one AIR, three kernels, one proof."""
    digests = []
    for extra in (dict(ZKHIP_FORCE_JIT="1", ZKHIP_JIT_TILE="1"), dict(ZKHIP_FORCE_JIT="1"), dict(ZKHIP_NO_JIT="1")):
        env = dict(os.environ, PYTHONPATH=ROOT, **extra)
        out = subprocess.run([sys.executable, "-c", TILED_SCRIPT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()[1])
    assert digests[0] == digests[1] == digests[2]
