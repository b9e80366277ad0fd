"""GPU: every kernel FORM a configuration field selects gives the default form's results (eight instantiations of the transform
pass, the forked constraint kernels and the 768-lane row sponge had no test).

* zkhip_config.ntt_log_lanes = 8 / 9: `k_ntt_pass4_ct<LOG_R, LOG_C, LOG_T>` with LOG_T = 8, 9 for every pass shape the dispatcher hands them
  (passes of 2^7 .. 2^10 rows: transforms of 2^11 .. 2^22 points and their coset extensions) -- forward, inverse, bit-reversed output and
  the LDE against the oracle's DFT (the definition: SURVEY.md A.2).
* zkhip_config.quot_streams = 1 .. 4: the compiled constraint kernels of a many-chip proof forked over side streams -- proof bytes equal to
  the one-stream proof (and to the oracle's for the small set).
* zkhip_config.hash_block = 768 (and the other multiples of 64 the range check admits): the row sponge of trees of >= 2^20 rows -- the
  root, sampled layers and a whole proof equal the 256-lane form's."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_lanes", [8, 9])
@pytest.mark.parametrize("log_n,width", [(11, 5), (12, 3), (13, 16), (14, 2), (15, 7), (16, 3), (17, 2), (18, 5), (19, 2), (20, 3), (21, 1), (22, 2)])
def test_transform_passes_of_256_and_512_lanes(zk, ora, log_lanes, log_n, width):
    zk.set_config(ntt_log_lanes=log_lanes)   # (restored after the test: tests/conftest.py)
    rng = np.random.default_rng(1000 * log_lanes + log_n)
    m = ora.rand_field(rng, (width, 1 << log_n))
    exp = ora.dft_batch(m, log_n)
    t = zk.upload(m.reshape(-1))
    zk.ntt_batch(t, log_n, width)
    assert (zk.download(t).reshape(width, -1) == exp).all()
    zk.ntt_batch(t, log_n, width, inverse=True)
    assert (zk.download(t).reshape(width, -1) == m).all()
    t2 = zk.upload(m.reshape(-1))
    zk.ntt_batch(t2, log_n, width, bitrev_out=True)
    idx = np.array([int(format(i, "0%db" % log_n)[::-1], 2) for i in range(1 << log_n)])
    assert (zk.download(t2).reshape(width, -1) == exp[:, idx]).all()
    if log_n <= 20:
        out = zk.lde_batch(zk.upload(m.reshape(-1)), log_n, 1, width, 31)
        assert (zk.download(out).reshape(width, -1) == ora.coset_lde_batch(m, log_n, 1, 31, bitrev_out=True)).all()


def test_forked_constraint_kernels_give_the_one_stream_proof(zk, ora):
    zk.set_config(jit=2)
    params = (1, 0, 8, 3, 4)
    airs = air.ChipSet(n_chips=6, log_max=8, log_min=3, total_width=60, seed=1, log_table=2).gen(1)
    exp = ora.stark_prove(params, airs).tobytes()
    pvs = [a["pvs"] for a in airs]
    for qs in (0, 1, 2, 3, 4):
        zk.set_config(jit=2, quot_streams=qs)
        pk = z.ProvingKey(zk, params, airs)
        d = [zk.upload(a["trace"].reshape(-1)) for a in airs]
        assert pk.prove(d, pvs) == exp, "quot_streams = %d" % qs
        assert pk.prove(d, pvs) == exp   # (the side streams are joined: the key's workspace is free for the next proof)
    # a chunk-circuit-sized set: 42 chips of 2^4 .. 2^14 rows, a dozen short compiled kernels per proof
    big = air.ChipSet(n_chips=42, log_max=14, log_min=4, total_width=300, seed=3, log_table=4).gen(2)
    bp = [a["pvs"] for a in big]
    proofs = []
    for qs in (0, 4, 2):
        zk.set_config(jit=2, quot_streams=qs)
        pk = z.ProvingKey(zk, z.DEFAULT_PARAMS, big)
        d = [zk.upload(a["trace"].reshape(-1)) for a in big]
        for _ in range(3):
            proofs.append(pk.prove(d, bp))
    assert len(set(proofs)) == 1
    assert z.verify(z.DEFAULT_PARAMS, pk.verifying_airs(), bp, proofs[0]) == 0


@pytest.mark.parametrize("hash_block", [768, 512, 64])
def test_row_sponge_workgroups_of_other_sizes(zk, ora, hash_block):
    log_n, width = 20, 37
    rng = np.random.default_rng(5)
    m = ora.rand_field(rng, (width, 1 << log_n))
    d = zk.upload(m.reshape(-1))
    ref = zk.merkle_commit([(d, log_n, width)])
    zk.set_config(hash_block=hash_block)
    got = zk.merkle_commit([(d, log_n, width)])
    assert got.root.tolist() == ref.root.tolist()
    assert got.root.tolist() == ora.Tree([m]).root.tolist()
    # a whole proof whose LDE trees have 2^20 rows
    sa = air.SyntheticAir(width=24, n_free=8, n_bool=2, n_boundary=2, seed=33)
    tr, pv = sa.gen_trace(19, seed=3)
    airs = [dict(program=sa.program(), log_height=19, width=24, n_pvs=len(pv))]
    dt = zk.upload(tr.reshape(-1))
    zk.set_config(hash_block=256)
    base = z.ProvingKey(zk, z.DEFAULT_PARAMS, airs).prove([dt], [pv])
    zk.set_config(hash_block=hash_block)
    assert z.ProvingKey(zk, z.DEFAULT_PARAMS, airs).prove([dt], [pv]) == base
    assert z.verify(z.DEFAULT_PARAMS, airs, [pv], base) == 0


def test_the_fused_middle_of_the_lde_is_bit_exact(zk, ora, monkeypatch):
    """ZKHIP_LDE_FUSED=1: inverse pass 2 and every coset's forward pass 1 in one kernel over one LDS tile (csrc/ntt.hip k_ntt_lde_fused; off
    by default -- measured no faster, profiles/round06_lde_fused.txt).  2^22 points: == the four-pass form's bytes, == the oracle on a
    column; both occupancy variants."""
    log_n, width = 22, 3
    rng = np.random.default_rng(22)
    m = ora.rand_field(rng, (width, 1 << log_n))
    d = zk.upload(m.reshape(-1))
    base = zk.download(zk.lde_batch(d, log_n, 1, width, 31))
    assert (base.reshape(width, -1)[1] == ora.coset_lde_batch(m[1:2], log_n, 1, 31, bitrev_out=True)[0]).all()
    for waves in ("8", "4"):
        monkeypatch.setenv("ZKHIP_LDE_FUSED", "1")
        monkeypatch.setenv("ZKHIP_LDE_FUSED_WAVES", waves)
        got = zk.download(zk.lde_batch(d, log_n, 1, width, 31))
        assert (got == base).all(), "fused LDE differs (waves %s)" % waves
        base4 = zk.download(zk.lde_batch(d, log_n, 2, width, 7))          # four cosets, another shift
        monkeypatch.delenv("ZKHIP_LDE_FUSED")
        assert (zk.download(zk.lde_batch(d, log_n, 2, width, 7)) == base4).all()


def test_the_shared_rows_constraint_kernel_gives_the_oracles_proof(tmp_path):
    """ZKHIP_JIT_SHARED=1 (csrc/quotient_jit.hpp: NW waves per 64 rows, wave w takes instances w, w + NW, ..., selectors from a table generated
    with the key; the default for chips of >= 2^20 LDE rows, forced here for small ones): chips of 2^12 rows with buses, preprocessed keys and a shared range table -- proof bytes == the
    oracle's.  A process of its own: the form is chosen when the kernel is generated, from the environment."""
    import os
    import subprocess
    import sys

    code = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_lib as ora
import zkvm_prover_amd as z
from zkvm_prover_amd import air
ora.lib()
zk = z.Context(0)
zk.set_config(jit=2)
params = (1, 0, 8, 3, 4)
airs = air.ChipSet(n_chips=5, log_max=12, log_min=11, total_width=70, seed=4, log_table=3).gen(1)
exp = ora.stark_prove(params, airs).tobytes()
pk = z.ProvingKey(zk, params, airs)
pvs = [a["pvs"] for a in airs]
got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], pvs)
assert got == exp, "shared-rows constraint kernel: proof differs from the oracle's"
assert z.verify(params, pk.verifying_airs(), pvs, got) == 0
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZKHIP_JIT_SHARED="1", ZKHIP_JIT_CACHE_DIR=str(tmp_path)), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    assert any(f.name.endswith(".hsaco") and b"#define NW " in f.read_bytes() for f in tmp_path.iterdir()), "no shared-rows kernel was generated"
