"""Trace builder for air.duplex_air from a transcript script -- a Python restatement of p3's DuplexChallenger (observe
overwrites the rate lanes in order and duplexes when eight are pending; sample duplexes if anything is pending or nothing is
left, then pops from the END of the output lanes).  Test infrastructure: the permutation is the oracle's."""
import numpy as np


def run_script(ora, script):
    """script: list of ("observe", [values]) / ("sample", n).  -> (rows [n_rows][50], io claims [(seq, lane, value, kind)], samples)"""
    state = np.zeros(16, np.uint32)
    pending, out_left = [], 0           # observed values not yet absorbed; output lanes not yet sampled
    rows, io, samples = [], [], []

    def duplex():
        nonlocal state, pending, out_left
        st_in = state.copy()
        st_in[:len(pending)] = pending
        st_out = ora.permute(st_in)
        seq = len(rows)
        for j, v in enumerate(pending):
            io.append((seq, j, int(v), 0))
        rows.append(dict(st_in=st_in, st_out=st_out, k=len(pending), sampled=0))
        state, pending, out_left = st_out.copy(), [], 8

    for op, arg in script:
        if op == "observe":
            for v in arg:
                pending.append(int(v))
                out_left = 0
                if len(pending) == 8:
                    duplex()
        else:
            for _ in range(arg):
                if pending or out_left == 0:
                    duplex()
                out_left -= 1
                samples.append(int(state[out_left]))
                io.append((len(rows) - 1, out_left, int(state[out_left]), 1))
                rows[-1]["sampled"] += 1
    tr = np.zeros((len(rows), 50), np.uint32)
    for r, row in enumerate(rows):
        tr[r, 0:16], tr[r, 16:32] = row["st_in"], row["st_out"]
        tr[r, 32:32 + row["k"]] = 1
        tr[r, 48 - row["sampled"]:48] = 1
        tr[r, 48], tr[r, 49] = r, 1
    return tr, io, samples


def padded(tr, log_height):
    out = np.zeros((tr.shape[1], 1 << log_height), np.uint32)
    out[:, :tr.shape[0]] = tr.T
    return out


def io_table(io, log_height):
    from collections import Counter

    cnt = Counter(io)
    t = np.zeros((5, 1 << log_height), np.uint32)
    for r, (c, m) in enumerate(sorted(cnt.items())):
        t[:4, r], t[4, r] = c, m
    return t
