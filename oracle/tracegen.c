/*
 * oracle/tracegen.c -- CPU restatements of the device-side trace generators.
 * TEST INFRASTRUCTURE.  PARITY UNPINNED (the chips' generators live in un-vendored OpenVM crates; this restates the
 * published structure: a range-checker chip's trace is the histogram of the values requested from it).
 * The Poseidon2 AIR generator is ora_poseidon2_air_trace in poseidon2.c (it needs that file's layers).
 */
#include "zk_oracle.h"

/* counts[v] (+)= #{ i : values[i] == v } mod p, canonical; returns the number of values >= 2^log_table (not counted) */
size_t ora_range_counts(const uint32_t *values, size_t n, unsigned log_table, uint32_t *counts, int accumulate) {
    const size_t T = (size_t)1 << log_table;
    size_t bad = 0;
    if (!accumulate)
        for (size_t v = 0; v < T; v++) counts[v] = 0;
    for (size_t i = 0; i < n; i++) {
        if (values[i] >= T) {
            bad++;
            continue;
        }
        counts[values[i]] = ora_add(counts[values[i]], 1);
    }
    return bad;
}

/* range-tuple table (OpenVM RangeTupleCheckerChip<2>): counts[x * size_y + y], canonical; returns the requests out of range */
size_t ora_range_tuple_counts(const uint32_t *xs, const uint32_t *ys, size_t n, uint32_t size_x, uint32_t size_y, uint32_t *counts,
                              int accumulate) {
    const size_t T = (size_t)size_x * size_y;
    size_t bad = 0;
    if (!accumulate)
        for (size_t v = 0; v < T; v++) counts[v] = 0;
    for (size_t i = 0; i < n; i++) {
        if (xs[i] >= size_x || ys[i] >= size_y) {
            bad++;
            continue;
        }
        uint32_t *c = &counts[(size_t)xs[i] * size_y + ys[i]];
        *c = ora_add(*c, 1);
    }
    return bad;
}

/* bitwise-operation lookup table (OpenVM BitwiseOperationLookupChip<bits>): trace = 2 columns of 2^(2 bits) rows,
 * row (x << bits) + y; column op (0 range, 1 xor) counts the requests */
size_t ora_bitwise_lookup_counts(const uint32_t *xs, const uint32_t *ys, const uint32_t *ops, size_t n, unsigned bits, uint32_t *trace,
                                 int accumulate) {
    const size_t R = (size_t)1 << (2 * bits);
    size_t bad = 0;
    if (!accumulate)
        for (size_t v = 0; v < 2 * R; v++) trace[v] = 0;
    for (size_t i = 0; i < n; i++) {
        if (xs[i] >> bits || ys[i] >> bits || ops[i] > 1) {
            bad++;
            continue;
        }
        uint32_t *c = &trace[ops[i] * R + ((size_t)xs[i] << bits) + ys[i]];
        *c = ora_add(*c, 1);
    }
    return bad;
}

/* volatile memory boundary chip (OpenVM VolatileBoundaryChip): rows sorted by (address space, pointer); 8 columns with
 * stride 2^log_height: as, ptr, initial, final, final timestamp, is_valid, gap_lo, gap_hi (gap = key_next - key - 1, key =
 * as * 2^ptr_bits + ptr, 16-bit low limb); returns the number of out-of-range or duplicated addresses */
#include <stdlib.h>
typedef struct {
    uint64_t key;
    uint32_t idx;
} ora_bkey;
static int ora_bkey_cmp(const void *a, const void *b) {
    const ora_bkey *x = (const ora_bkey *)a, *y = (const ora_bkey *)b;
    return x->key < y->key ? -1 : x->key > y->key ? 1 : (x->idx < y->idx ? -1 : x->idx > y->idx);
}
size_t ora_memory_boundary_trace(const uint32_t *as, const uint32_t *ptr, const uint32_t *init, const uint32_t *fin, const uint32_t *ts,
                                 size_t n, unsigned as_bits, unsigned ptr_bits, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    ora_bkey *k = (ora_bkey *)malloc((n + 1) * sizeof(ora_bkey));
    for (size_t i = 0; i < n; i++) {
        if ((as_bits < 32 && (as[i] >> as_bits)) || (ptr_bits < 32 && (ptr[i] >> ptr_bits))) bad++;
        k[i].key = ((uint64_t)as[i] << 32) | ptr[i];
        k[i].idx = (uint32_t)i;
    }
    qsort(k, n, sizeof(ora_bkey), ora_bkey_cmp);
    for (size_t q = 0; q < 8; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n; r++) {
        const uint32_t j = k[r].idx;
        trace[0 * N + r] = (uint32_t)(k[r].key >> 32) % ORA_P;
        trace[1 * N + r] = (uint32_t)k[r].key % ORA_P;
        trace[2 * N + r] = init[j];
        trace[3 * N + r] = fin[j];
        trace[4 * N + r] = ts[j] % ORA_P;
        trace[5 * N + r] = 1;
        if (r + 1 < n) {
            if (k[r + 1].key == k[r].key) bad++;
            const uint64_t a = ((k[r].key >> 32) << ptr_bits) + (uint32_t)k[r].key, b = ((k[r + 1].key >> 32) << ptr_bits) + (uint32_t)k[r + 1].key;
            const uint64_t gap = b - a - 1;
            trace[6 * N + r] = (uint32_t)(gap & 0xffffu);
            trace[7 * N + r] = (uint32_t)((gap >> 16) % ORA_P);
        }
    }
    free(k);
    return bad;
}

/* RV32 base ALU core chip (OpenVM BaseAluCoreAir) from records (opcode 0..4 = add, sub, xor, or, and; operands b, c): 18 columns with
 * stride 2^log_height: a[4] | b[4] | c[4] | 5 opcode flags | is_valid; xor_counts[(x << 8) | y] (+)= the bitwise-lookup requests of
 * the rows ((b_i, c_i) for bitwise opcodes, (a_i, a_i) for add / sub), canonical mod p.  Returns the number of bad opcodes. */
size_t ora_rv32_alu_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                          uint32_t *xor_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 18; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r], b = bs[r], c = cs[r];
        if (op > 4) {
            bad++;
            continue;
        }
        const uint32_t a = op == 0 ? b + c : op == 1 ? b - c : op == 2 ? (b ^ c) : op == 3 ? (b | c) : (b & c);
        for (int i = 0; i < 4; i++) {
            const uint32_t ai = (a >> (8 * i)) & 255u, bi = (b >> (8 * i)) & 255u, ci = (c >> (8 * i)) & 255u;
            trace[(size_t)i * N + r] = ai, trace[(size_t)(4 + i) * N + r] = bi, trace[(size_t)(8 + i) * N + r] = ci;
            const uint32_t x = op >= 2 ? bi : ai, y = op >= 2 ? ci : ai;
            xor_counts[(x << 8) | y] = ora_add(xor_counts[(x << 8) | y], 1);
        }
        trace[(size_t)(12 + op) * N + r] = 1;
        trace[(size_t)17 * N + r] = 1;
    }
    return bad;
}

/* RV32 multiplication core chip (OpenVM MultiplicationCoreAir): 13 columns a[4] | b[4] | c[4] | is_valid; tuple_counts[a_i * size_y +
 * carry_i] (+)= 1 for the four (limb, carry) pairs of every record */
void ora_rv32_mul_trace(const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace, uint32_t *tuple_counts,
                        uint32_t size_y) {
    const size_t N = (size_t)1 << log_height;
    for (size_t q = 0; q < 13; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        uint32_t bl[4], cl[4], carry = 0;
        for (int i = 0; i < 4; i++) bl[i] = (bs[r] >> (8 * i)) & 255u, cl[i] = (cs[r] >> (8 * i)) & 255u;
        for (int i = 0; i < 4; i++) {
            uint32_t acc = carry;
            for (int k = 0; k <= i; k++) acc += bl[k] * cl[i - k];
            const uint32_t ai = acc & 255u;
            carry = acc >> 8;
            trace[(size_t)i * N + r] = ai, trace[(size_t)(4 + i) * N + r] = bl[i], trace[(size_t)(8 + i) * N + r] = cl[i];
            uint32_t *t = &tuple_counts[(size_t)ai * size_y + carry];
            *t = ora_add(*t, 1);
        }
        trace[(size_t)12 * N + r] = 1;
    }
}

/* Program chip / execution frames (zkhip_program_freq_tracegen, zkhip_exec_frame_tracegen): freq[k] = how often instruction k was
 * executed; frame row r = the 9 program fields of instruction idx[r] and is_valid.  Canonical values.  Returns the number of
 * indices that are not rows of the program. */
size_t ora_program_freq_trace(const uint32_t *idx, size_t n, unsigned log_height, uint32_t *freq) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t r = 0; r < N; r++) freq[r] = 0;
    for (size_t i = 0; i < n; i++) {
        if (idx[i] >= N) bad++;
        else freq[idx[i]] = ora_add(freq[idx[i]], 1);
    }
    return bad;
}
size_t ora_exec_frame_trace(const uint32_t *idx, size_t n, const uint32_t *program, size_t n_program, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 10; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        if (idx[r] >= n_program) {
            bad++;
            continue;
        }
        for (size_t q = 0; q < 9; q++) trace[q * N + r] = program[q * n_program + idx[r]];
        trace[9 * N + r] = 1;
    }
    return bad;
}

/* RV32 less-than core (zkhip_rv32_lt_tracegen): 18 columns b[4] | c[4] | cmp | is_slt is_sltu | b_msb_f c_msb_f | marker[4] | diff_val,
 * canonical; range_counts[(x << 8) | y] (+)= 1 for the row's range requests.  Returns the number of records with opcode > 1. */
size_t ora_rv32_lt_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                         uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 18; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r];
        if (op > 1) {
            bad++;
            continue;
        }
        long bl[4], cl[4];
        for (int i = 0; i < 4; i++) bl[i] = (bs[r] >> (8 * i)) & 255u, cl[i] = (cs[r] >> (8 * i)) & 255u;
        const int is_slt = op == 0;
        /* the operands as integers, most significant limb signed for SLT */
        const long bm = is_slt && bl[3] >= 128 ? bl[3] - 256 : bl[3], cm = is_slt && cl[3] >= 128 ? cl[3] - 256 : cl[3];
        long bv = bm, cv = cm;
        for (int i = 2; i >= 0; i--) bv = bv * 256 + bl[i], cv = cv * 256 + cl[i];
        int mark = -1;
        for (int i = 3; i >= 0 && mark < 0; i--)
            if ((i == 3 ? bm : bl[i]) != (i == 3 ? cm : cl[i])) mark = i;
        for (int i = 0; i < 4; i++) trace[(size_t)i * N + r] = (uint32_t)bl[i], trace[(size_t)(4 + i) * N + r] = (uint32_t)cl[i];
        trace[(size_t)8 * N + r] = bv < cv;
        trace[(size_t)(9 + op) * N + r] = 1;
        trace[(size_t)11 * N + r] = (uint32_t)(bm < 0 ? (long)ORA_P + bm : bm);
        trace[(size_t)12 * N + r] = (uint32_t)(cm < 0 ? (long)ORA_P + cm : cm);
        const long sh = is_slt ? 128 : 0;
        uint32_t *t = &range_counts[((bm + sh) << 8) | (cm + sh)];
        *t = ora_add(*t, 1);
        if (mark >= 0) {
            long d = (mark == 3 ? cm - bm : cl[mark] - bl[mark]);
            if (d < 0) d = -d;
            trace[(size_t)(13 + mark) * N + r] = 1;
            trace[(size_t)17 * N + r] = (uint32_t)d;
            t = &range_counts[(d - 1) << 8];
            *t = ora_add(*t, 1);
        }
    }
    return bad;
}

/* Memory access chip (zkhip_memory_access_tracegen): 10 columns as | ptr | prev_data | prev_ts | data | ts | is_read | is_valid | gap_lo |
 * gap_hi, canonical.  Returns the number of refused records. */
size_t ora_memory_access_trace(const uint32_t *as, const uint32_t *ptr, const uint32_t *prev_data, const uint32_t *prev_ts, const uint32_t *data,
                               const uint32_t *ts, const uint32_t *is_read, size_t n, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 10; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        if (as[r] >= ORA_P || ptr[r] >= ORA_P || prev_data[r] > 0xffff || data[r] > 0xffff || ts[r] >= ORA_P || prev_ts[r] >= ts[r] || is_read[r] > 1 ||
            (is_read[r] && data[r] != prev_data[r])) {
            bad++;
            continue;
        }
        const uint32_t gap = ts[r] - prev_ts[r] - 1;
        const uint32_t row[10] = {as[r], ptr[r], prev_data[r], prev_ts[r], data[r], ts[r], is_read[r], 1, gap & 0xffff, gap >> 16};
        for (size_t q = 0; q < 10; q++) trace[q * N + r] = row[q];
    }
    return bad;
}

/* RV32 shift core (zkhip_rv32_shift_tracegen): 32 columns a[4] | b[4] | c0 | is_sll is_srl is_sra | bit_marker[8] | limb_marker[4] |
 * carry[4] | sign | q | mult_left | mult_right, canonical; range_counts / xor_counts[(x << 8) | y] (+)= 1 per lookup request.
 * The result limbs come from a limb-wise restatement of the shift (not from the C shift operators the device kernel uses). */
size_t ora_rv32_shift_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                            uint32_t *range_counts, uint32_t *xor_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 32; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r];
        if (op > 2) {
            bad++;
            continue;
        }
        const uint32_t s = cs[r] & 31u, bsh = s & 7u, lsh = s >> 3, c0 = cs[r] & 255u, q = c0 >> 5, mult = 1u << bsh;
        uint32_t bl[4], t[4], cy[4], al[4];
        for (int i = 0; i < 4; i++) bl[i] = (bs[r] >> (8 * i)) & 255u;
        const uint32_t sign = op == 2 ? bl[3] >> 7 : 0;
        if (op == 0) { /* bit shift limb by limb towards the top, then move up by whole limbs */
            uint32_t carry = 0;
            for (int k = 0; k < 4; k++) {
                const uint32_t v = bl[k] * mult + carry;
                t[k] = v & 255u, carry = cy[k] = v >> 8;
            }
            for (int i = 0; i < 4; i++) al[i] = (uint32_t)i < lsh ? 0 : t[i - lsh];
        } else { /* towards the bottom: limb k keeps its high bits and takes the bits falling out of limb k + 1 (or the sign fill) */
            for (int k = 0; k < 4; k++) cy[k] = bl[k] & (mult - 1u);
            for (int k = 0; k < 4; k++) {
                const uint32_t in = k == 3 ? sign * (mult - 1u) : cy[k + 1];
                t[k] = (bl[k] >> bsh) + in * (256u / mult);
            }
            for (int i = 0; i < 4; i++) al[i] = i + lsh > 3 ? 255u * sign : t[i + lsh];
        }
        uint32_t row[32] = {0};
        for (int i = 0; i < 4; i++) row[i] = al[i], row[4 + i] = bl[i], row[24 + i] = cy[i];
        row[8] = c0, row[9 + op] = 1, row[12 + bsh] = 1, row[20 + lsh] = 1, row[28] = sign, row[29] = q, row[op == 0 ? 30 : 31] = mult;
        for (size_t c = 0; c < 32; c++) trace[c * N + r] = row[c];
        uint32_t *p;
        for (int i = 0; i < 4; i++) p = &range_counts[(cy[i] << 8) | (mult - 1u - cy[i])], *p = ora_add(*p, 1);
        p = &range_counts[(al[0] << 8) | al[1]], *p = ora_add(*p, 1);
        p = &range_counts[(al[2] << 8) | al[3]], *p = ora_add(*p, 1);
        p = &range_counts[(q << 8) | (32u * q)], *p = ora_add(*p, 1);
        if (op == 2) p = &xor_counts[(bl[3] << 8) | 128u], *p = ora_add(*p, 1);
    }
    return bad;
}

/* RV32 branch-equal core (zkhip_rv32_branch_eq_tracegen): 17 columns a[4] | b[4] | taken | imm | is_beq is_bne | diff_inv_marker[4] |
 * pc_inc, canonical; imm is the offset as a canonical field element.  Returns the number of refused records. */
size_t ora_rv32_branch_eq_trace(const uint32_t *opc, const uint32_t *as, const uint32_t *bs, const uint32_t *imms, size_t n, unsigned log_height,
                                uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 17; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r];
        if (op > 1 || imms[r] >= ORA_P) {
            bad++;
            continue;
        }
        const int equal = as[r] == bs[r], taken = op == 0 ? equal : !equal;
        int first = -1;
        for (int i = 0; i < 4; i++) {
            const uint32_t ai = (as[r] >> (8 * i)) & 255u, bi = (bs[r] >> (8 * i)) & 255u;
            trace[(size_t)i * N + r] = ai, trace[(size_t)(4 + i) * N + r] = bi;
            if (first < 0 && ai != bi) first = i, trace[(size_t)(12 + i) * N + r] = ora_inv(ora_sub(ai, bi));
        }
        trace[(size_t)8 * N + r] = (uint32_t)taken;
        trace[(size_t)9 * N + r] = imms[r];
        trace[(size_t)(10 + op) * N + r] = 1;
        trace[(size_t)16 * N + r] = taken ? imms[r] : 4;
    }
    return bad;
}

/* RV32 branch-less-than core (zkhip_rv32_branch_lt_tracegen): 23 columns a[4] | b[4] | cmp_lt | taken | imm | is_blt is_bltu is_bge is_bgeu |
 * a_msb_f b_msb_f | marker[4] | diff_val | pc_inc, canonical; range_counts as in ora_rv32_lt_trace.  cmp_lt comes from the operands
 * as integers (signed for BLT / BGE), the marker scan from the limbs. */
size_t ora_rv32_branch_lt_trace(const uint32_t *opc, const uint32_t *as, const uint32_t *bs, const uint32_t *imms, size_t n, unsigned log_height,
                                uint32_t *trace, uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 23; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r];
        if (op > 3 || imms[r] >= ORA_P) {
            bad++;
            continue;
        }
        const int is_signed = (op & 1) == 0, is_ge = op >= 2;
        long al[4], bl[4];
        for (int i = 0; i < 4; i++) al[i] = (as[r] >> (8 * i)) & 255u, bl[i] = (bs[r] >> (8 * i)) & 255u;
        const long am = is_signed && al[3] >= 128 ? al[3] - 256 : al[3], bm = is_signed && bl[3] >= 128 ? bl[3] - 256 : bl[3];
        long av = am, bv = bm;
        for (int i = 2; i >= 0; i--) av = av * 256 + al[i], bv = bv * 256 + bl[i];
        const int lt = av < bv, taken = lt != is_ge;
        int mark = -1;
        for (int i = 3; i >= 0 && mark < 0; i--)
            if ((i == 3 ? am : al[i]) != (i == 3 ? bm : bl[i])) mark = i;
        for (int i = 0; i < 4; i++) trace[(size_t)i * N + r] = (uint32_t)al[i], trace[(size_t)(4 + i) * N + r] = (uint32_t)bl[i];
        trace[(size_t)8 * N + r] = (uint32_t)lt, trace[(size_t)9 * N + r] = (uint32_t)taken, trace[(size_t)10 * N + r] = imms[r];
        trace[(size_t)(11 + op) * N + r] = 1;
        trace[(size_t)15 * N + r] = (uint32_t)(am < 0 ? (long)ORA_P + am : am);
        trace[(size_t)16 * N + r] = (uint32_t)(bm < 0 ? (long)ORA_P + bm : bm);
        trace[(size_t)22 * N + r] = taken ? imms[r] : 4;
        const long sh = is_signed ? 128 : 0;
        uint32_t *t = &range_counts[((am + sh) << 8) | (bm + sh)];
        *t = ora_add(*t, 1);
        if (mark >= 0) {
            long d = mark == 3 ? bm - am : bl[mark] - al[mark];
            if (d < 0) d = -d;
            trace[(size_t)(17 + mark) * N + r] = 1;
            trace[(size_t)21 * N + r] = (uint32_t)d;
            t = &range_counts[(d - 1) << 8];
            *t = ora_add(*t, 1);
        }
    }
    return bad;
}

/* RV32 JAL / LUI core (zkhip_rv32_jal_lui_tracegen): 9 columns pc | imm | rd[4] | is_jal is_lui | pc_inc, canonical.  Record = (opcode
 * 0 = JAL, 1 = LUI; pc; imm = the offset as a field element for JAL, the 20-bit immediate for LUI).  Range requests as the AIR
 * sends them: (rd0, rd1), (rd2, rd3), and (4 rd3, 0) for JAL. */
static void ora_bump(uint32_t *range_counts, uint32_t x, uint32_t y) {
    uint32_t *t = &range_counts[(x << 8) | y];
    *t = ora_add(*t, 1);
}
size_t ora_rv32_jal_lui_trace(const uint32_t *opc, const uint32_t *pcs, const uint32_t *imms, size_t n, unsigned log_height, uint32_t *trace,
                              uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 9; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r], pc = pcs[r], imm = imms[r];
        if (op > 1 || (op == 0 ? (imm >= ORA_P || pc >= (1u << 30) - 4) : (imm >> 20) != 0) || pc >= ORA_P) {
            bad++;
            continue;
        }
        const uint32_t rd = op == 0 ? pc + 4 : imm << 12;
        trace[0 * N + r] = pc, trace[1 * N + r] = imm;
        for (int i = 0; i < 4; i++) trace[(size_t)(2 + i) * N + r] = (rd >> (8 * i)) & 255u;
        trace[(size_t)(6 + op) * N + r] = 1;
        trace[8 * N + r] = op == 0 ? imm : 4;
        ora_bump(range_counts, rd & 255u, (rd >> 8) & 255u);
        ora_bump(range_counts, (rd >> 16) & 255u, rd >> 24);
        if (op == 0) ora_bump(range_counts, (rd >> 24) * 4, 0);
    }
    return bad;
}

/* RV32 AUIPC core (zkhip_rv32_auipc_tracegen): 14 columns pc | imm | pc_limb[4] | imm_limb[3] | rd[4] | is_valid; record = (pc, the
 * 20-bit immediate); rd = pc + (imm << 12) mod 2^32 from 64-bit integer arithmetic. */
size_t ora_rv32_auipc_trace(const uint32_t *pcs, const uint32_t *imms, size_t n, unsigned log_height, uint32_t *trace, uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 14; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t pc = pcs[r], imm = imms[r];
        if (pc >= (1u << 30) || (imm >> 20) != 0) {
            bad++;
            continue;
        }
        const uint32_t rd = (uint32_t)(((uint64_t)pc + ((uint64_t)imm << 12)) & 0xffffffffu), im16 = imm << 4;
        uint32_t pl[4], il[3], dl[4];
        for (int i = 0; i < 4; i++) pl[i] = (pc >> (8 * i)) & 255u, dl[i] = (rd >> (8 * i)) & 255u;
        for (int i = 0; i < 3; i++) il[i] = (im16 >> (8 * i)) & 255u;
        trace[0 * N + r] = pc, trace[1 * N + r] = imm;
        for (int i = 0; i < 4; i++) trace[(size_t)(2 + i) * N + r] = pl[i], trace[(size_t)(9 + i) * N + r] = dl[i];
        for (int i = 0; i < 3; i++) trace[(size_t)(6 + i) * N + r] = il[i];
        trace[13 * N + r] = 1;
        ora_bump(range_counts, pl[0], pl[1]), ora_bump(range_counts, pl[2], 4 * pl[3]), ora_bump(range_counts, il[0], il[1]);
        ora_bump(range_counts, il[2], dl[1]), ora_bump(range_counts, dl[2], dl[3]);
    }
    return bad;
}

/* RV32 JALR core (zkhip_rv32_jalr_tracegen): 20 columns pc | imm | imm_limb[2] | imm_sign | rs1[4] | rd[4] | t[4] | lsb | to_pc | is_valid;
 * record = (pc, rs1, the raw 12-bit immediate).  t = rs1 + sext(imm) from integer arithmetic, to_pc = t & ~1, rd = pc + 4. */
size_t ora_rv32_jalr_trace(const uint32_t *pcs, const uint32_t *rs1s, const uint32_t *imms, size_t n, unsigned log_height, uint32_t *trace,
                           uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 20; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t pc = pcs[r], rs1 = rs1s[r], imm = imms[r];
        if (pc >= (1u << 30) - 4 || (imm >> 12) != 0) {
            bad++;
            continue;
        }
        const uint32_t sign = imm >> 11, ext = sign ? imm | 0xfffff000u : imm, t = rs1 + ext, rd = pc + 4, to_pc = t & ~1u;
        if (to_pc >= (1u << 30)) {   /* program counters are 30-bit values (the top limb of t is range-checked as 4 t_3) */
            bad++;
            continue;
        }
        trace[0 * N + r] = pc, trace[1 * N + r] = imm, trace[2 * N + r] = imm & 255u, trace[3 * N + r] = imm >> 8, trace[4 * N + r] = sign;
        for (int i = 0; i < 4; i++) {
            trace[(size_t)(5 + i) * N + r] = (rs1 >> (8 * i)) & 255u;
            trace[(size_t)(9 + i) * N + r] = (rd >> (8 * i)) & 255u;
            trace[(size_t)(13 + i) * N + r] = (t >> (8 * i)) & 255u;
        }
        trace[17 * N + r] = t & 1u, trace[18 * N + r] = to_pc, trace[19 * N + r] = 1;
        ora_bump(range_counts, imm & 255u, ((imm >> 8) - 8 * sign) * 32);
        ora_bump(range_counts, (t & 255u) >> 1, (t >> 8) & 255u);
        ora_bump(range_counts, (t >> 16) & 255u, (t >> 24) * 4);
        ora_bump(range_counts, rd & 255u, (rd >> 8) & 255u);
        ora_bump(range_counts, (rd >> 16) & 255u, (rd >> 24) * 4);
    }
    return bad;
}

/* RV32 high-multiplication core (zkhip_rv32_mulh_tracegen): 21 columns a[4] | b[4] | c[4] | a_mul[4] | b_sign c_sign | is_mulh is_mulhsu
 * is_mulhu, canonical; record = (opcode 0 = MULH, 1 = MULHSU, 2 = MULHU; operands).  a and a_mul are the halves of the 64-bit
 * product from INTEGER arithmetic; the carries sent to the tuple table come from the limb sums the AIR states, and the function
 * returns the number of records (there must be none) where the two disagree or a carry leaves the table. */
size_t ora_rv32_mulh_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                           uint32_t *tuple_counts, uint32_t size_y, uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 21; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r], bv = bs[r], cv = cs[r];
        if (op > 2) {
            bad++;
            continue;
        }
        const uint32_t b_sign = op != 2 ? bv >> 31 : 0, c_sign = op == 0 ? cv >> 31 : 0;
        const int64_t sb = b_sign ? (int64_t)(int32_t)bv : (int64_t)bv, sc = c_sign ? (int64_t)(int32_t)cv : (int64_t)cv;
        const uint64_t prod = (uint64_t)(sb * sc);   /* |sb|, |sc| < 2^32 and not both 2^32 - 1 signed: fits */
        uint32_t l[8], m[8];
        for (int i = 0; i < 4; i++) l[i] = (bv >> (8 * i)) & 255u, m[i] = (cv >> (8 * i)) & 255u;
        for (int i = 4; i < 8; i++) l[i] = b_sign * 255u, m[i] = c_sign * 255u;
        uint32_t carry = 0;
        int ok = 1;
        for (int i = 0; i < 8; i++) {
            uint32_t acc = carry;
            for (int k = 0; k <= i; k++) acc += l[k] * m[i - k];
            const uint32_t limb = acc & 255u;
            carry = acc >> 8;
            if (limb != ((prod >> (8 * i)) & 255u) || carry >= size_y) ok = 0;
            trace[(size_t)(i < 4 ? 12 + i : i - 4) * N + r] = limb;
            if (carry < size_y) {
                uint32_t *t = &tuple_counts[(size_t)limb * size_y + carry];
                *t = ora_add(*t, 1);
            }
        }
        if (!ok) bad++;
        for (int i = 0; i < 4; i++) trace[(size_t)(4 + i) * N + r] = l[i], trace[(size_t)(8 + i) * N + r] = m[i];
        trace[(size_t)16 * N + r] = b_sign, trace[(size_t)17 * N + r] = c_sign;
        trace[(size_t)(18 + op) * N + r] = 1;
        if (op != 2) ora_bump(range_counts, 2 * (l[3] - 128 * b_sign), 0);
        if (op == 0) ora_bump(range_counts, 2 * (m[3] - 128 * c_sign), 0);
    }
    return bad;
}

/* RV32 load/store cores (zkhip_rv32_loadstore_tracegen): 33 columns read[4] | prev[4] | write[4] | case flag[20] | sign, canonical.
 * Record = (case 0..19 in the order LW, LHU@0 LHU@2, LBU@0..3, SW, SH@0 SH@2, SB@0..3, LH@0 LH@2, LB@0..3; read word; prev word).
 * write comes from integer shifts and masks of the words (loads extract and extend, stores merge into prev). */
size_t ora_rv32_loadstore_trace(const uint32_t *cases, const uint32_t *reads, const uint32_t *prevs, size_t n, unsigned log_height, uint32_t *trace,
                                uint32_t *range_counts) {
    static const uint8_t SHIFT[20] = {0, 0, 2, 0, 1, 2, 3, 0, 0, 2, 0, 1, 2, 3, 0, 2, 0, 1, 2, 3};
    static const uint8_t BYTES[20] = {4, 2, 2, 1, 1, 1, 1, 4, 2, 2, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1};
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 33; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t cs = cases[r], rd = reads[r], pv = prevs[r];
        if (cs > 19) {
            bad++;
            continue;
        }
        const unsigned sh = 8u * SHIFT[cs], nb = BYTES[cs];
        const uint32_t mask = nb == 4 ? 0xffffffffu : (1u << (8 * nb)) - 1;
        const int is_store = cs >= 7 && cs <= 13, is_signed = cs >= 14;
        uint32_t wr, sign = 0;
        if (is_store) {
            wr = (pv & ~(mask << sh)) | ((rd & mask) << sh);
        } else {
            wr = (rd >> sh) & mask;
            if (is_signed) {
                sign = (wr >> (8 * nb - 1)) & 1u;
                if (sign) wr |= ~mask;
                ora_bump(range_counts, 2 * (((wr >> (8 * (nb - 1))) & 255u) - 128 * sign), 0);
            }
        }
        for (int i = 0; i < 4; i++) {
            trace[(size_t)i * N + r] = (rd >> (8 * i)) & 255u;
            trace[(size_t)(4 + i) * N + r] = (pv >> (8 * i)) & 255u;
            trace[(size_t)(8 + i) * N + r] = (wr >> (8 * i)) & 255u;
        }
        trace[(size_t)(12 + cs) * N + r] = 1;
        trace[(size_t)32 * N + r] = sign;
    }
    return bad;
}

/* RV32 division core (zkhip_rv32_divrem_tracegen): 41 columns b[4] | c[4] | q[4] | r[4] | c_abs[4] | r_abs[4] | b_sign c_sign q_sign r_sign |
 * k_c k_r | zero_divisor c_sum_inv | marker[4] | diff | is_div is_divu is_rem is_remu, canonical.  Record = (opcode 0 = DIV, 1 = DIVU,
 * 2 = REM, 3 = REMU; dividend b; divisor c).  q and r come from C's integer division with RISC-V's two exceptions (c = 0: q = all
 * ones, r = b; DIV/REM of -2^31 by -1: q = -2^31, r = 0); the carries sent to the tuple table come from the limb sums the AIR
 * states.  Returns the number of records (there must be none) with an opcode > 3, a limb sum that is not a multiple of 256 or a
 * carry outside the table. */
size_t ora_rv32_divrem_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace,
                             uint32_t *tuple_counts, uint32_t size_y, uint32_t *range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 41; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t row = 0; row < n && row < N; row++) {
        const uint32_t op = opc[row], b = bs[row], c = cs[row];
        if (op > 3) {
            bad++;
            continue;
        }
        const int is_signed = (op & 1) == 0;
        uint32_t q, r;
        int overflow = 0;
        if (c == 0) {
            q = 0xffffffffu, r = b;
        } else if (is_signed && b == 0x80000000u && c == 0xffffffffu) {
            q = b, r = 0, overflow = 1;
        } else if (is_signed) {
            q = (uint32_t)((int32_t)b / (int32_t)c), r = (uint32_t)((int32_t)b % (int32_t)c);
        } else {
            q = b / c, r = b % c;
        }
        const uint32_t b_sign = is_signed ? b >> 31 : 0, c_sign = is_signed ? c >> 31 : 0, r_sign = is_signed ? r >> 31 : 0;
        const uint32_t q_sign = is_signed && !overflow ? q >> 31 : 0;
        const uint32_t ca = c_sign ? 0u - c : c, ra = r_sign ? 0u - r : r;
        const uint32_t kc = c_sign && (c & 0xffffu) ? 1 : 0, kr = r_sign && (r & 0xffffu) ? 1 : 0;
        uint32_t w[4] = {b, c, q, r};
        for (int g = 0; g < 4; g++)
            for (int i = 0; i < 4; i++) trace[(size_t)(4 * g + i) * N + row] = (w[g] >> (8 * i)) & 255u;
        for (int i = 0; i < 4; i++) trace[(size_t)(16 + i) * N + row] = (ca >> (8 * i)) & 255u, trace[(size_t)(20 + i) * N + row] = (ra >> (8 * i)) & 255u;
        trace[(size_t)24 * N + row] = b_sign, trace[(size_t)25 * N + row] = c_sign, trace[(size_t)26 * N + row] = q_sign, trace[(size_t)27 * N + row] = r_sign;
        trace[(size_t)28 * N + row] = kc, trace[(size_t)29 * N + row] = kr;
        trace[(size_t)30 * N + row] = c == 0;
        uint32_t csum = 0;
        for (int i = 0; i < 4; i++) csum += (c >> (8 * i)) & 255u;
        trace[(size_t)31 * N + row] = c == 0 ? 0 : ora_inv(csum);
        trace[(size_t)(37 + op) * N + row] = 1;
        int ok = 1;
        if (c != 0) {   /* |r| < |c|: the most significant differing limb of the magnitudes */
            int mark = -1;
            for (int i = 3; i >= 0 && mark < 0; i--)
                if (((ca >> (8 * i)) & 255u) != ((ra >> (8 * i)) & 255u)) mark = i;
            if (mark < 0 || ra >= ca) {
                ok = 0;
            } else {
                const uint32_t d = ((ca >> (8 * mark)) & 255u) - ((ra >> (8 * mark)) & 255u);
                trace[(size_t)(32 + mark) * N + row] = 1, trace[(size_t)36 * N + row] = d;
                ora_bump(range_counts, d - 1, 0);
            }
        }
        /* c q + r - b over eight sign-extended limbs */
        int64_t l[8], m[8], rr[8], bb[8], carry = 0;
        for (int i = 0; i < 8; i++) {
            l[i] = i < 4 ? (c >> (8 * i)) & 255u : 255 * c_sign, m[i] = i < 4 ? (q >> (8 * i)) & 255u : 255 * q_sign;
            rr[i] = i < 4 ? (r >> (8 * i)) & 255u : 255 * r_sign, bb[i] = i < 4 ? (b >> (8 * i)) & 255u : 255 * b_sign;
        }
        for (int i = 0; i < 8; i++) {
            int64_t acc = carry + rr[i] - bb[i];
            for (int k = 0; k <= i; k++) acc += l[k] * m[i - k];
            if (acc < 0 || (acc & 255) != 0 || (acc >> 8) >= size_y) {
                ok = 0;
                break;
            }
            carry = acc >> 8;
            const uint32_t limb = (uint32_t)(i < 4 ? m[i] : rr[i - 4]);
            uint32_t *t = &tuple_counts[(size_t)limb * size_y + (uint32_t)carry];
            *t = ora_add(*t, 1);
        }
        if (!ok) bad++;
        if (is_signed) ora_bump(range_counts, 2 * (((b >> 24) & 255u) - 128 * b_sign), 2 * (((c >> 24) & 255u) - 128 * c_sign));
        ora_bump(range_counts, ca & 255u, (ca >> 8) & 255u), ora_bump(range_counts, (ca >> 16) & 255u, ca >> 24);
        ora_bump(range_counts, ra & 255u, (ra >> 8) & 255u), ora_bump(range_counts, (ra >> 16) & 255u, ra >> 24);
    }
    return bad;
}

/* MMCS path chip (zkhip_mmcs_path_tracegen; air.py mmcs_path_air): 39 columns root[8] | parent[8] | a[8] | b[8] | bit | is_inj | is_first |
 * is_last | is_real | idx | lvl, canonical.  Path p: leaf digest leaf[8 p ..], index index[p], steps path_start[p] .. path_start[p+1]
 * BOTTOM-UP (step_kind 0 = sibling digest, consuming the next index bit from the least significant end; 1 = injected row digest
 * of shorter matrices; step_digest[8 s ..]).  The digests are compressed bottom-up with ora_compress, the rows written top-down
 * (row path_start[p] is the compression that yields the root).  hash_inputs[16 r ..] = a || b of row r (what the Poseidon2 chip
 * must permute), claims[18 c ..] = (root, lvl, idx, digest) in row order (injections, then the leaf of each path).  Returns the
 * number of paths that do not fit the trace or do not end in a sibling step. */
size_t ora_mmcs_path_trace(const uint32_t *leaf, const uint32_t *index, const uint32_t *path_start, const uint32_t *step_kind,
                           const uint32_t *step_digest, size_t n_paths, unsigned log_height, uint32_t *trace, uint32_t *hash_inputs,
                           uint32_t *claims, size_t *n_claims) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0, nc = 0;
    for (size_t q = 0; q < 39; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t p = 0; p < n_paths; p++) {
        const size_t s0 = path_start[p], s1 = path_start[p + 1], ns = s1 - s0;
        if (s1 > N || ns == 0 || ns > 64 || step_kind[s0] != 0) {
            bad++;
            continue;
        }
        uint32_t a[64][8], b[64][8], par[64][8], bits[64], node[8];
        unsigned sibs = 0;
        for (int i = 0; i < 8; i++) node[i] = leaf[8 * p + i];
        for (size_t j = 0; j < ns; j++) {
            const uint32_t *d = &step_digest[8 * (s0 + j)];
            bits[j] = step_kind[s0 + j] == 0 ? (index[p] >> sibs) & 1u : 0;
            if (step_kind[s0 + j] == 0) sibs++;
            for (int i = 0; i < 8; i++) a[j][i] = bits[j] ? d[i] : node[i], b[j][i] = bits[j] ? node[i] : d[i];
            ora_compress(a[j], b[j], par[j]);
            for (int i = 0; i < 8; i++) node[i] = par[j][i];
        }
        uint32_t idx = 0, lvl = 0;
        for (size_t k = 0; k < ns; k++) {   /* top-down */
            const size_t j = ns - 1 - k, r = s0 + k;
            const uint32_t inj = step_kind[s0 + j] != 0;
            idx = idx * (2 - inj) + bits[j], lvl += 1 - inj;
            for (int i = 0; i < 8; i++) {
                trace[(size_t)i * N + r] = node[i], trace[(size_t)(8 + i) * N + r] = par[j][i];
                trace[(size_t)(16 + i) * N + r] = a[j][i], trace[(size_t)(24 + i) * N + r] = b[j][i];
                hash_inputs[16 * r + i] = a[j][i], hash_inputs[16 * r + 8 + i] = b[j][i];
            }
            trace[(size_t)32 * N + r] = bits[j], trace[(size_t)33 * N + r] = inj, trace[(size_t)34 * N + r] = k == 0, trace[(size_t)35 * N + r] = j == 0;
            trace[(size_t)36 * N + r] = 1, trace[(size_t)37 * N + r] = idx, trace[(size_t)38 * N + r] = lvl;
            if (inj || j == 0) {
                uint32_t *c = &claims[18 * nc++];
                for (int i = 0; i < 8; i++) c[i] = node[i], c[10 + i] = inj ? b[j][i] : (bits[j] ? b[j][i] : a[j][i]);
                c[8] = lvl, c[9] = idx;
            }
        }
    }
    *n_claims = nc;
    return bad;
}

/* Native field-arithmetic core (zkhip_field_arith_tracegen): 8 columns a | b | c | is_add is_sub is_mul is_div | divisor_inv, canonical.
 * Record = (opcode 0 = ADD, 1 = SUB, 2 = MUL, 3 = DIV; b; c) field elements.  Returns the number of bad records (opcode > 3,
 * operand >= p, division by zero). */
size_t ora_field_arith_trace(const uint32_t *opc, const uint32_t *bs, const uint32_t *cs, size_t n, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 8; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r], b = bs[r], c = cs[r];
        if (op > 3 || b >= ORA_P || c >= ORA_P || (op == 3 && c == 0)) {
            bad++;
            continue;
        }
        const uint32_t inv = op == 3 ? ora_inv(c) : 0;
        trace[0 * N + r] = op == 0 ? ora_add(b, c) : op == 1 ? ora_sub(b, c) : op == 2 ? ora_mul(b, c) : ora_mul(b, inv);
        trace[1 * N + r] = b, trace[2 * N + r] = c, trace[(size_t)(3 + op) * N + r] = 1, trace[7 * N + r] = inv;
    }
    return bad;
}

/* Native field-extension core (zkhip_field_ext_tracegen): 20 columns x[4] | y[4] | z[4] | is_add is_sub is_mul is_div | divisor_inv[4];
 * record = (opcode 0..3; x[4]; y[4]) with xs / ys holding four words per record. */
size_t ora_field_ext_trace(const uint32_t *opc, const uint32_t *xs, const uint32_t *ys, size_t n, unsigned log_height, uint32_t *trace) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 20; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t op = opc[r], *x = xs + 4 * r, *y = ys + 4 * r;
        int ok = op <= 3;
        for (int i = 0; i < 4; i++) ok = ok && x[i] < ORA_P && y[i] < ORA_P;
        if (ok && op == 3 && !(y[0] | y[1] | y[2] | y[3])) ok = 0;
        if (!ok) {
            bad++;
            continue;
        }
        uint32_t z[4], inv[4] = {0, 0, 0, 0};
        if (op == 0) for (int i = 0; i < 4; i++) z[i] = ora_add(x[i], y[i]);
        else if (op == 1) for (int i = 0; i < 4; i++) z[i] = ora_sub(x[i], y[i]);
        else if (op == 2) ora_ext_mul(x, y, z);
        else ora_ext_inv(y, inv), ora_ext_mul(x, inv, z);
        for (int i = 0; i < 4; i++)
            trace[(size_t)i * N + r] = x[i], trace[(size_t)(4 + i) * N + r] = y[i], trace[(size_t)(8 + i) * N + r] = z[i], trace[(size_t)(16 + i) * N + r] = inv[i];
        trace[(size_t)(12 + op) * N + r] = 1;
    }
    return bad;
}

/* Variable range checker multiplicities (zkhip_var_range_counts_tracegen): counts[2^bits - 1 + value] (+)= 1 per request
 * (value, bits), bits <= max_bits, value < 2^bits; bits == NULL means const_bits for every request.  Canonical inputs.
 * Returns the number of requests outside the table. */
size_t ora_var_range_counts(const uint32_t *values, const uint32_t *bits, uint32_t const_bits, size_t n, unsigned max_bits, uint32_t *counts) {
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t b = bits ? bits[i] : const_bits, v = values[i];
        if (b > max_bits || (b < 32 && v >= (1u << b))) {
            bad++;
            continue;
        }
        uint32_t *t = &counts[((size_t)1 << b) - 1 + v];
        *t = ora_add(*t, 1);
    }
    return bad;
}

/* Native CASTF core (zkhip_castf_tracegen): 6 columns x | limb[4] | is_valid, canonical; record = x < 2^30.  The four limb checks
 * are counted into the variable range checker's table (max_bits >= 8): counts[2^bits - 1 + limb] with bits 8, 8, 8, 6. */
size_t ora_castf_trace(const uint32_t *xs, size_t n, unsigned log_height, uint32_t *trace, uint32_t *var_range_counts) {
    const size_t N = (size_t)1 << log_height;
    size_t bad = 0;
    for (size_t q = 0; q < 6; q++)
        for (size_t r = 0; r < N; r++) trace[q * N + r] = 0;
    for (size_t r = 0; r < n && r < N; r++) {
        const uint32_t x = xs[r];
        if (x >> 30) {
            bad++;
            continue;
        }
        trace[r] = x, trace[5 * N + r] = 1;
        for (int i = 0; i < 4; i++) {
            const uint32_t limb = (x >> (8 * i)) & 255u, bits = i < 3 ? 8 : 6;
            trace[(size_t)(1 + i) * N + r] = limb;
            uint32_t *t = &var_range_counts[(1u << bits) - 1 + limb];
            *t = ora_add(*t, 1);
        }
    }
    return bad;
}
