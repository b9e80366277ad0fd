"""GPU: device-side trace generation of the Poseidon2 AIR (zkhip_poseidon2_air_tracegen) -- the trace equals the
oracle's cell for cell, proofs made from the device-resident trace equal the oracle's byte for byte, and at a size
the oracle does not reach the trace is checked through the permutation kernel and sampled rows of the AIR."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air
import oracle_lib as O

pytestmark = pytest.mark.gpu
NOPV = np.zeros(0, np.uint32)


@pytest.mark.parametrize("log_n,n_perms", [(0, 1), (0, 0), (3, 8), (6, 50), (9, 512), (12, 4000), (10, 0)])
def test_device_trace_equals_oracle(zk, log_n, n_perms):
    rng = np.random.default_rng(100 + log_n)
    inputs = O.rand_field(rng, (n_perms, 16))
    exp = O.poseidon2_air_trace(inputs, log_n)
    d_in = zk.upload(inputs.reshape(-1)) if n_perms else zk.upload(np.zeros(16, np.uint32))[:0]
    d_tr = zk.poseidon2_air_tracegen(d_in, log_n)
    got = zk.download(d_tr).reshape(298, 1 << log_n)
    assert (got == exp).all()


@pytest.mark.parametrize("jit", ["interpreter", "jit"])
def test_proof_from_device_trace_equals_oracle(zk, ora, jit, monkeypatch):
    zk.set_config(jit=0 if jit == "interpreter" else 2)   # (restored after the test: tests/conftest.py)
    params = (1, 0, 8, 3, 4)
    rng = np.random.default_rng(5)
    inputs = O.rand_field(rng, (200, 16))
    tr = O.poseidon2_air_trace(inputs, 8)
    ftr, fpv = air.fibonacci_trace(5)
    airs = [dict(program=air.poseidon2_air().program(), log_height=8, width=298, n_pvs=0, trace=tr, pvs=NOPV),
            dict(program=air.fibonacci_air().program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
    pvs = [a["pvs"] for a in airs]
    exp = ora.stark_prove(params, airs)
    pk = z.ProvingKey(zk, params, airs)
    d_traces = [zk.poseidon2_air_tracegen(zk.upload(inputs.reshape(-1)), 8), zk.upload(ftr.reshape(-1))]
    zk.profile_reset()
    zk.profile_enable(True)
    got = pk.prove(d_traces, pvs)
    zk.profile_enable(False)
    assert got == exp.tobytes()
    assert z.verify(params, airs, pvs, got) == 0
    # the Poseidon2 AIR does not compress into shape classes: the straight-line form of the whole DAG must have been
    # compiled (no silent fall-back to the interpreter)
    labels = zk.profile_read()
    assert ("quotient_eval_jit" in labels) == (jit == "jit"), sorted(labels)


def test_large_device_trace_properties_and_proof(zk):
    import torch

    log_n = 16
    n = (1 << log_n) - 77
    rng = np.random.default_rng(8)
    inputs = O.rand_field(rng, (n, 16))
    d_in = zk.upload(inputs.reshape(-1))
    d_tr = zk.poseidon2_air_tracegen(d_in, log_n)
    tr = d_tr.view(298, 1 << log_n)
    # inputs and outputs: the first / last 16 columns against the batch permutation kernel
    d_st = torch.zeros((1 << log_n, 16), dtype=torch.int32, device=d_in.device)
    d_st[:n] = d_in.view(n, 16)
    assert torch.equal(tr[:16].t().contiguous(), d_st)
    zk.poseidon2_permute_batch(d_st, 1 << log_n)
    zk.sync()
    assert torch.equal(tr[-16:].t().contiguous(), d_st)
    # sampled rows satisfy every constraint (rows are independent: no transition constraints)
    rows = np.concatenate([[0, n - 1, n, (1 << log_n) - 1], rng.integers(0, 1 << log_n, 12)])
    sample = zk.download(tr[:, torch.from_numpy(rows).to(d_in.device)].contiguous().view(-1)).reshape(298, len(rows))
    prog = air.poseidon2_air().program()
    assert air.check_trace(prog, sample, NOPV) == []
    params = z.DEFAULT_PARAMS
    airs = [dict(program=prog, log_height=log_n, width=298, n_pvs=0)]
    pk = z.ProvingKey(zk, params, airs)
    proof = pk.prove([d_tr], [NOPV])
    assert z.verify(params, airs, [NOPV], proof) == 0


@pytest.mark.parametrize("jit", ["interpreter", "jit"])
def test_poseidon2_chip_on_a_bus_device_trace_equals_oracle(zk, ora, jit, monkeypatch):
    """Requester + Poseidon2 chip over a 24-field bus: the chip's 298 permutation columns come from the device
    generator, the multiplicity column is the caller's; proof bytes equal the oracle's."""
    import torch
    from test_p2air_cpu import hasher_pair

    zk.set_config(jit=0 if jit == "interpreter" else 2)   # (restored after the test: tests/conftest.py)
    params = (1, 0, 8, 3, 4)
    airs, uniq = hasher_pair(log_user=6, n_req=40, log_chip=7, seed=11)
    pvs = [NOPV, NOPV]
    exp = ora.stark_prove(params, airs)
    assert ora.stark_verify(params, airs, exp) == 0
    N = 1 << 7
    d_chip = torch.empty(299 * N, dtype=torch.int32, device="cuda:0")
    zk.poseidon2_air_tracegen(zk.upload(uniq.reshape(-1)), 7, d_chip)
    d_chip[298 * N:] = zk.upload(airs[1]["trace"][298])
    assert (zk.download(d_chip).reshape(299, N) == airs[1]["trace"]).all()
    pk = z.ProvingKey(zk, params, airs)
    got = pk.prove([zk.upload(airs[0]["trace"].reshape(-1)), d_chip], pvs)
    assert got == exp.tobytes()
    assert z.verify(params, airs, pvs, got) == 0


@pytest.mark.parametrize("log_table,n", [(0, 5), (4, 1000), (13, 100000), (14, 100000), (20, 300000), (6, 0)])
def test_range_counts_tracegen_equals_oracle(zk, log_table, n):
    rng = np.random.default_rng(log_table)
    vals = rng.integers(0, 1 << log_table, n).astype(np.uint32)
    if n > 10:
        vals[: n // 3] = vals[0]                      # a heavy hitter (atomic contention)
    exp, bad = O.range_counts(vals, log_table)
    assert bad == 0
    d_vals = zk.upload(vals) if n else zk.upload(np.zeros(1, np.uint32))[:0]
    d_c = zk.range_counts_tracegen(d_vals, log_table)
    assert (zk.download(d_c) == exp).all()
    # accumulate a second requesting column
    more = rng.integers(0, 1 << log_table, 777).astype(np.uint32)
    exp2, _ = O.range_counts(more, log_table, counts=exp)
    zk.range_counts_tracegen(zk.upload(more), log_table, d_c, accumulate=True)
    assert (zk.download(d_c) == exp2).all()


def test_range_counts_rejects_out_of_range_values(zk):
    vals = np.array([1, 2, 3, 16, 5], np.uint32)
    with pytest.raises(Exception, match="outside the table"):
        zk.range_counts_tracegen(zk.upload(vals), 4)


def test_range_table_trace_from_device_counts_proves(zk, ora):
    """The range-table chip's trace generated on the device from the user chip's resident column: proof == oracle."""
    params = (1, 0, 8, 3, 4)
    u, m, prep = air.range_traces(9, 6, seed=5)
    airs = [dict(program=air.range_user_air().program(), log_height=9, width=4, n_pvs=0, trace=u, pvs=NOPV),
            dict(program=air.range_table_air().program(), log_height=6, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep)]
    exp = ora.stark_prove(params, airs)
    d_user = zk.upload(u.reshape(-1))
    d_table = zk.range_counts_tracegen(d_user[: 1 << 9], 6)      # column 0 of the user trace
    pk = z.ProvingKey(zk, params, airs)
    got = pk.prove([d_user, d_table], [NOPV, NOPV])
    assert got == exp.tobytes()


def test_all_cpp_hasher_demo_proves_and_rejects_tampering():
    """tools/hasher_demo.cpp: AIRs from include/zkhip_air.hpp, device tracegen, keygen / prove / verify through the
    C ABI only -- no Python in the loop."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zkvm-prover_amd", "hasher_demo")
    assert os.path.exists(exe), "hasher_demo not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.run([exe, "10"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "verified" in out.stdout and "tampered digest: rejected" in out.stdout
