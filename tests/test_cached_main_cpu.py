"""CPU: cached main partitions (OpenVM-v1 `cached_mains` -- the reference's stored proofs carry two main-trace commitments:
one cached partition and the common main).  The first `cached_width` main columns of an AIR are committed in a tree of their
own; the oracle prover produces such proofs, both verifiers (oracle C, product host) accept them and reject every tampered
word; the layout table knows the cached commitments."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

PARAMS = (1, 0, 10, 3, 4)
NOPV = np.zeros(0, np.uint32)


def cached_case():
    sa = air.SyntheticAir(width=12, n_free=5, n_bool=2, n_boundary=2, seed=3)
    sa.builder.cached_width = 5
    tr, pv = sa.gen_trace(6, seed=1)
    ftr, fpv = air.fibonacci_trace(4)
    fb = air.fibonacci_air()
    fb.cached_width = 1
    s, t = air.lookup_traces(6, 4, seed=1)
    lb = air.lookup_sender_air()
    lb.cached_width = 2
    u, m, prep = air.range_traces(5, 3, seed=2)
    return [dict(program=sa.program(), log_height=6, width=12, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=fb.program(), log_height=4, width=2, n_pvs=3, trace=ftr, pvs=fpv),
            dict(program=lb.program(), log_height=6, width=3, n_pvs=0, trace=s, pvs=NOPV),
            dict(program=air.lookup_table_air().program(), log_height=4, width=3, n_pvs=0, trace=t, pvs=NOPV),
            dict(program=air.range_user_air().program(), log_height=5, width=4, n_pvs=0, trace=u, pvs=NOPV),
            dict(program=air.range_table_air().program(), log_height=3, width=1, n_pvs=0, trace=m, pvs=NOPV, prep=prep)]


def verifying(ora, airs):
    out = []
    for a in airs:
        v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs")}
        if a.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(PARAMS, a)
        out.append(v)
    return out


def test_cached_partitions_prove_verify_and_bind(ora):
    airs = cached_case()
    pvs = [a["pvs"] for a in airs]
    proof = ora.stark_prove(PARAMS, airs)
    assert int(proof[0]) == 0x31504B5A + 1 + 2 + 4      # LogUp + preprocessed + cached
    vk = verifying(ora, airs)
    assert ora.stark_verify(PARAMS, airs, proof) == 0
    assert z.verify(PARAMS, vk, pvs, proof.tobytes()) == 0
    lay = z.proof_layout(PARAMS, vk)
    assert lay["n_cached"] == 3 and lay["roots_cached"] == lay["root_main"] + 8 and lay["n_words"] == len(proof)
    # the cached commitment is the stand-alone commitment of those columns' LDE
    cw = 5
    lde = ora.coset_lde_batch(airs[0]["trace"][:cw], 6, 1, 31)
    assert ora.Tree([lde]).root.tolist() == proof[lay["roots_cached"]:lay["roots_cached"] + 8].tolist()
    rng = np.random.default_rng(0)
    spots = [1, 5, 13, 14, 21, 22, 29, 30, 37, 38, len(proof) - 1] + rng.integers(0, len(proof), 100).tolist()
    for pos in sorted(set(spots)):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % ora.P
        assert ora.stark_verify(PARAMS, airs, bad) != 0, "oracle accepted tampered word %d" % pos
        assert z.verify(PARAMS, vk, pvs, bad.tobytes()) != 0, "product accepted tampered word %d" % pos


def test_cached_and_plain_programs_are_different_statements(ora):
    airs = cached_case()
    plain = [dict(a) for a in airs]
    sa = air.SyntheticAir(width=12, n_free=5, n_bool=2, n_boundary=2, seed=3)
    plain[0]["program"] = sa.program()
    p_cached, p_plain = ora.stark_prove(PARAMS, airs), ora.stark_prove(PARAMS, plain)
    assert len(p_cached) != len(p_plain)
    assert ora.stark_verify(PARAMS, airs, p_plain) != 0 and ora.stark_verify(PARAMS, plain, p_cached) != 0
    pvs = [a["pvs"] for a in airs]
    assert z.verify(PARAMS, verifying(ora, plain), pvs, p_cached.tobytes()) != 0


def test_cached_width_must_leave_a_common_part():
    b = air.AirBuilder(2, 0)
    b.assert_zero(b.var(0) - b.var(1))
    prog = b.program().tolist() + [air.CACHED_MAGIC, 2]
    a = [dict(program=np.array(prog, dtype=np.uint32), log_height=3, width=2, n_pvs=0)]
    assert z.verify(PARAMS, a, [NOPV], np.zeros(64, np.uint32).tobytes()) == -3


def test_cached_proof_into_the_reference_container(ora):
    """With a cached partition the v1 container carries the commitments as the reference's stored proofs do:
    main_trace = [cached..., common]; the conversion is lossless."""
    import refproof_v1 as rp

    params = (1, 0, 10, 0, 4)
    airs = cached_case()
    pvs = [a["pvs"] for a in airs]
    vk = []
    for a in airs:
        v = {k: a[k] for k in ("program", "log_height", "width", "n_pvs")}
        if a.get("prep") is not None:
            v["prep_commit"] = ora.prep_commit(params, a)
        vk.append(v)
    proof = ora.stark_prove(params, airs).tobytes()
    v1 = z.proof_to_v1(params, vk, pvs, proof)
    p = rp.decode_proofs((1).to_bytes(8, "little") + v1)[0]
    lay = z.proof_layout(params, vk)
    words = np.frombuffer(proof, dtype=np.uint32)
    assert len(p["main_trace"]) == 4   # three cached partitions + the common main, like the reference's [cached, common]
    assert [rp.from_monty(x) for x in p["main_trace"][3]] == words[lay["root_main"]:lay["root_main"] + 8].tolist()
    assert [rp.from_monty(x) for x in p["main_trace"][0]] == words[lay["roots_cached"]:lay["roots_cached"] + 8].tolist()
    q0 = p["fri"]["query_proofs"][0]["input_proof"]
    # batches: preprocessed tree, 3 cached trees (widths 5, 1, 2), common main (6 matrices), after-challenge, quotient
    assert [[len(r) for r in b["opened_values"]] for b in q0[:5]] == [[1], [5], [1], [2], [7, 1, 1, 3, 4, 1]]
    assert [len(m) for m in p["opened"]["main"]] == [1, 1, 1, 6]
    s = z.proof_decode_v1(v1, z.V1_SINGLE)
    assert s["n_main_commits"] == 4 and s["n_preprocessed"] == 1
    back, pvs_back = z.proof_from_v1(params, vk, v1)
    assert back == proof and z.verify(params, vk, pvs_back, back) == 0
