"""GPU: byte parity with the oracle AT THE REFERENCE'S PARAMETERS (crates/circuits/chunk-circuit/openvm.toml:1-6: blow-up 2, 100 queries,
16 + 16 bits of proof of work) for the proofs of the guest flow -- every other oracle-parity test of a segment or node proof runs at
PARAMS = (1, 0, 4, 3, 3) to keep the (slow, obviously-right) oracle in seconds.  oracle/fast refuses AIRs with preprocessed traces or bus
interactions, which a segment and a node circuit both have, so the checker here is oracle/stark.c and the instances are sized for it:
  * one VM segment (22 chips, frames of 2^13 rows, device-generated traces): HIP proof == oracle proof, 100 queries, PoW 16 + 16;
  * one 4-child leaf node of the aggregation tree over children proven at the same parameters (the verifier circuit replays 4 x 100
    queries and the proofs of work in-circuit): device traces == twins, HIP node proof == oracle proof."""
import numpy as np
import pytest

import zkvm_prover_amd as z

import recursion_util as ru
import rv32_model as rv
import vm2_util as v2
from test_gpu_vm2 import PC_BASE, device_traces
from test_vm_cpu import fib_program

pytestmark = pytest.mark.gpu
REF = (1, 0, 100, 16, 16)
NOPV = ru.NOPV


def test_segment_proof_at_reference_parameters(zk, ora, tmp_path):
    words = fib_program()
    info, H, segs, image_root, pv_open = v2.dump_segments(tmp_path, rv.exe_bytes(words), (3000).to_bytes(4, "little"), 13)
    rec = segs[0]
    inst = v2.segment_instance(rec, words, PC_BASE, H)
    assert len(inst) == 22 and max(a["log_height"] for a in inst) >= 13
    prog = v2.program_table(words, PC_BASE, H[0])
    T = device_traces(zk, rec, prog, H)
    for a, d in enumerate(inst):
        assert (zk.download(T[a]).reshape(d["width"], -1) == d["trace"]).all(), "chip %d" % a
    pk = z.ProvingKey(zk, REF, inst)
    pvs = [d["pvs"] for d in inst]
    proof = pk.prove(T, pvs)
    assert z.verify(REF, pk.verifying_airs(), pvs, proof) == 0
    assert z.verify((1, 0, 99, 16, 16), pk.verifying_airs(), pvs, proof) != 0
    want = ora.stark_prove(REF, inst, cap_words=1 << 26).tobytes()
    assert len(proof) == len(want) and proof == want
    pk.close()


def test_leaf_node_proof_at_reference_parameters(zk, ora):
    kids = [ru.counter_segment(s, seed=i) for i, s in enumerate([3, 10, 17, 24])]
    vk = ru.verifying(REF, kids[0])
    pk = z.ProvingKey(zk, REF, kids[0])
    proofs = []
    for k in kids:
        proofs.append(pk.prove([zk.upload(a["trace"].reshape(-1)) for a in k], [a["pvs"] for a in k]))
        assert proofs[-1] == ora.stark_prove(REF, k).tobytes()
    pvs = [[a["pvs"] for a in k] for k in kids]
    rc = z.RecursionCircuit(REF, vk, 4, stmt=ru.COUNTER_STMT, uniform=True)
    st, npv = rc.witness(proofs, pvs)
    assert st == 0, rc.last_error()
    assert int(npv[8]) == 3 and int(npv[9]) == 31
    node = ru.node_instance(rc, npv)
    d_traces = rc.tracegen(zk)
    for a, d in zip(node, d_traces):
        assert (zk.download(d).reshape(a["width"], -1) == a["trace"]).all()
    npk = z.ProvingKey(zk, REF, node)
    proof = npk.prove(d_traces, [NOPV, NOPV, npv])
    assert z.verify(REF, npk.verifying_airs(), [NOPV, NOPV, npv], proof) == 0
    want = ora.stark_prove(REF, node, cap_words=1 << 26).tobytes()
    assert len(proof) == len(want) and proof == want
    # a child proven with another grinding witness / tampered query answer has no witness at these parameters either
    bad = bytearray(proofs[1])
    bad[len(bad) - 40] ^= 1
    assert rc.witness([proofs[0], bytes(bad)] + proofs[2:], pvs)[0] == -7
