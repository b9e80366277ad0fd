"""CPU: the transcript chip (air.duplex_air) against the oracle's challenger: a random script of observations and samples,
restated as duplexing rows, yields the oracle challenger's samples and satisfies the AIR; broken carry-over, a gap in the
overwritten prefix or a sample taken from the front are caught."""
import numpy as np

from zkvm_prover_amd import air
import duplex_util as du

NOPV = np.zeros(0, np.uint32)
P = 2013265921


def random_script(rng, n_ops):
    script = []
    for _ in range(n_ops):
        if rng.random() < 0.6:
            script.append(("observe", rng.integers(0, P, int(rng.integers(1, 20))).astype(np.uint32).tolist()))
        else:
            script.append(("sample", int(rng.integers(1, 12))))
    return script


def test_duplex_rows_reproduce_the_oracle_challenger(ora):
    rng = np.random.default_rng(3)
    script = random_script(rng, 60)
    tr, io, samples = du.run_script(ora, script)
    ch = ora.Challenger()
    want = []
    for op, arg in script:
        if op == "observe":
            ch.observe(np.array(arg, np.uint32))
        else:
            want += ch.sample(arg).tolist()
    assert samples == want and len(tr) > 40
    lh = int(np.ceil(np.log2(len(tr) + 1)))
    t = du.padded(tr, lh)
    prog = air.duplex_air(9, 10).program()
    assert air.check_trace(prog, t, NOPV) == [] and air.quotient_chunks(prog) == 2
    r = next(i for i in range(1, len(tr)) if tr[i][32] == 1 and tr[i][39] == 0)     # a row with a partial overwrite
    k = int(tr[r][32:40].sum())
    for col, row in ((8 + 3, 5), (k, r), (48, 7), (49, len(tr) - 1)):                 # capacity carry-over, rate carry-over, seq, is_real
        w = t.copy()
        w[col][row] = (int(w[col][row]) + 1) % P
        assert air.check_trace(prog, w, NOPV) != [], (col, row)
    w = t.copy()
    w[32 + k - 1][r], w[32 + k][r] = 0, 1                                            # a hole in the overwritten prefix
    assert air.check_trace(prog, w, NOPV) != []
    rs = next(i for i in range(len(tr)) if tr[i][47] == 1 and tr[i][40] == 0)
    w = t.copy()
    w[47][rs], w[40][rs] = 0, 1                                                      # sampled from the front instead of the end
    assert air.check_trace(prog, w, NOPV) != []
