"""CPU: the wrapped witnesses of ADVICE round 2 against the offline memory-checking AIRs (air.memory_access_air, air.memory_boundary_air):
a timestamp that runs backwards (prev_ts = 10, ts = 3 with gap limbs 65529 / 30719) and two boundary rows with ONE key (gap limbs
0 / 30720) satisfy the polynomial constraints modulo p -- what refuses them is the lookup of 8 gap_hi in the 16-bit range table
(gap_hi < 2^13: the gap stays below 2^29 and cannot stand for a negative difference)."""
import numpy as np

from zkvm_prover_amd import air

import vm2_util as v2

P = 2013265921
NOPV = np.zeros(0, np.uint32)


def _range_instance(chip_air, width, trace, log_height):
    """the chip + a 16-bit range table whose multiplicities are whatever the chip's in-range requests need"""
    inst = [dict(program=chip_air.program(), log_height=log_height, width=width, n_pvs=0, trace=trace, pvs=NOPV)]
    vals, ints = v2._eval_nodes(inst[0]["program"], trace, NOPV, None)
    cnt = np.zeros(1 << 16, np.int64)
    outside = 0
    for bus, sign, count, fields in ints:
        if bus != 5:
            continue
        for c, f in zip(vals[count], vals[fields[0]]):
            if c:
                if f < (1 << 16):
                    cnt[int(f)] += int(c)
                else:
                    outside += 1
    inst.append(dict(program=air.range_table_air(5).program(), log_height=16, width=1, n_pvs=0, trace=(cnt % P).astype(np.uint32).reshape(1, -1), pvs=NOPV,
                     prep=np.arange(1 << 16, dtype=np.uint32).reshape(1, -1)))
    return inst, outside


def test_a_timestamp_cannot_run_backwards():
    # columns: as ptr prev_data prev_ts data ts is_read is_valid gap_lo gap_hi
    honest = np.zeros((10, 2), np.uint32)
    honest[:, 0] = [1, 4, 7, 3, 7, 10, 1, 1, 6, 0]
    wrapped = honest.copy()
    gap = (3 - 10 - 1) % P
    wrapped[:, 0] = [1, 4, 7, 10, 7, 3, 1, 1, gap & 0xFFFF, gap >> 16]
    assert (int(wrapped[8, 0]), int(wrapped[9, 0])) == (65529, 30719)
    prog = air.memory_access_air().program()
    assert air.check_trace(prog, honest, NOPV) == [] and air.check_trace(prog, wrapped, NOPV) == []   # the polynomial identity holds modulo p
    inst, outside = _range_instance(air.memory_access_air(), 10, honest, 1)
    assert outside == 0 and 5 not in v2.bus_imbalance(inst)
    inst, outside = _range_instance(air.memory_access_air(), 10, wrapped, 1)
    assert outside == 1 and 5 in v2.bus_imbalance(inst)                                                # 8 * 30719 is not in the table


def test_two_boundary_rows_cannot_carry_one_key():
    # columns: as ptr initial final final_ts is_valid gap_lo gap_hi; key = as * 2^27 + ptr
    honest = np.zeros((8, 2), np.uint32)
    honest[:, 0] = [1, 5, 0, 9, 4, 1, 2, 0]
    honest[:, 1] = [1, 8, 0, 9, 4, 1, 0, 0]
    wrapped = honest.copy()
    gap = (0 - 1) % P                                   # key' - key - 1 with key' = key
    wrapped[1, 1] = 5
    wrapped[6, 0], wrapped[7, 0] = gap & 0xFFFF, gap >> 16
    assert (int(wrapped[6, 0]), int(wrapped[7, 0])) == (0, 30720)
    prog = air.memory_boundary_air().program()
    assert air.check_trace(prog, honest, NOPV) == [] and air.check_trace(prog, wrapped, NOPV) == []
    inst, outside = _range_instance(air.memory_boundary_air(), 8, honest, 1)
    assert outside == 0 and 5 not in v2.bus_imbalance(inst)
    inst, outside = _range_instance(air.memory_boundary_air(), 8, wrapped, 1)
    assert outside >= 1 and 5 in v2.bus_imbalance(inst)
