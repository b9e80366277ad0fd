"""GPU: a chunk-circuit-shaped AIR set (many chips of mixed heights, per-chip buses with compound messages, a shared
range table with preprocessed keys).  Small instance: proof bytes equal the oracle's.  42 chips (the reference's
chunk circuit has 42 OpenVM chips, AGENTS.md:183-185): verifies against the verifying key, tampering is rejected."""
import numpy as np
import pytest

import zkvm_prover_amd as z
from zkvm_prover_amd import air

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("jit", ["interpreter", "jit"])
def test_small_chipset_equals_oracle(zk, ora, jit, monkeypatch):
    zk.set_config(jit=0 if jit == "interpreter" else 2)   # (restored after the test: tests/conftest.py)
    params = (1, 0, 8, 3, 4)
    airs = air.ChipSet(n_chips=6, log_max=8, log_min=3, total_width=60, seed=1, log_table=2).gen(1)
    for a in airs:
        assert air.check_trace(a["program"], a["trace"], a["pvs"], a.get("prep")) == []
    exp = ora.stark_prove(params, airs)
    pk = z.ProvingKey(zk, params, airs)
    pvs = [a["pvs"] for a in airs]
    got = pk.prove([zk.upload(a["trace"].reshape(-1)) for a in airs], pvs)
    assert got == exp.tobytes()
    assert z.verify(params, pk.verifying_airs(), pvs, got) == 0


def test_42_chips_verify(zk):
    params = z.DEFAULT_PARAMS
    cs = air.ChipSet(n_chips=42, log_max=14, log_min=4, total_width=300, seed=3, log_table=4)
    airs = cs.gen(2)
    assert len(airs) == 43 and 250 <= sum(cs.widths) <= 350
    pk = z.ProvingKey(zk, params, airs)
    pvs = [a["pvs"] for a in airs]
    d_traces = [zk.upload(a["trace"].reshape(-1)) for a in airs]
    got = pk.prove(d_traces, pvs)
    vk = pk.verifying_airs()
    assert z.verify(params, vk, pvs, got) == 0
    assert pk.prove(d_traces, pvs) == got
    words = np.frombuffer(got, dtype=np.uint32).copy()
    rng = np.random.default_rng(5)
    for pos in [4, 13, 21, 40] + rng.integers(0, len(words), 12).tolist():
        bad = words.copy()
        bad[pos] = (int(bad[pos]) + 1) % air.P
        assert z.verify(params, vk, pvs, bad.tobytes()) != 0
    # one chip lies about a boolean it sent to the range table
    bad_airs = [dict(a) for a in airs]
    t = bad_airs[7]["trace"].copy()
    t[1, 0] ^= 1
    bad_airs[7]["trace"] = t
    d_bad = list(d_traces)
    d_bad[7] = zk.upload(t.reshape(-1))
    assert z.verify(params, vk, pvs, pk.prove(d_bad, pvs)) != 0
