"""Tiny full-STARK invocation used by __graft_entry__.smoke(): HIP proof bytes vs the oracle."""
import numpy as np


def run(ctx):
    import oracle_lib as ora
    import zkvm_prover_amd as z
    from zkvm_prover_amd import air

    params = (1, 0, 8, 4, 4)
    sa = air.SyntheticAir(width=24, n_free=8, n_bool=4, n_boundary=3, seed=1)
    tr, pv = sa.gen_trace(6, seed=2)
    fa = air.fibonacci_air()
    ftr, fpv = air.fibonacci_trace(5)
    airs = [dict(program=sa.program(), log_height=6, width=24, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=fa.program(), log_height=5, width=2, n_pvs=3, trace=ftr, pvs=fpv)]
    pk = z.ProvingKey(ctx, params, airs)
    proof = pk.prove([ctx.upload(a["trace"].reshape(-1)) for a in airs], [a["pvs"] for a in airs])
    exp = ora.stark_prove(params, airs)
    assert proof == exp.tobytes(), "STARK proof bytes differ from the oracle"
    assert z.verify(params, airs, [a["pvs"] for a in airs], proof) == 0
    print("smoke ok: full STARK proof (%d bytes) bit-exact vs oracle and verified" % len(proof))
