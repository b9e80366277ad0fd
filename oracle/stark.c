/*
 * oracle/stark.c -- CPU restatement of the whole proving path and its verifier.
 * TEST INFRASTRUCTURE (see zk_oracle.h).  Primitives pinned to the reference (zk_oracle.h); the protocol flow of this
 * file (transcript order, quotient, reduced openings, proof layout) is PARITY UNPINNED.
 *
 * What it restates: the STARK the reference obtains from `sdk.prove(..)`
 * (crates/prover/src/prover/mod.rs:355-357) in the shape BASELINE.json's
 * north_star names -- trace commit -> coset LDE -> Merkle-Poseidon2 commit ->
 * [bus interactions: LogUp after-challenge trace + commit] ->
 * constraint/quotient evaluation -> FRI fold loop with PoW and queries -- i.e.
 * the published Plonky3 uni-stark + TwoAdicFriPcs pipeline (multi-matrix,
 * mixed heights, OpenVM-v1-style LogUp phase) with the FRI parameters of
 * crates/circuits/chunk-circuit/openvm.toml:1-6.  The exact transcript / proof
 * layout of the pinned OpenVM v2 backend (openvm-stark-backend 2.0.0,
 * Cargo.lock:5337) is NOT available offline (SURVEY.md finding 2, 8c); the
 * layout below is this repository's own (DESIGN.md "Protocol") and is what the
 * HIP prover must reproduce byte for byte.
 *
 * AIR bytecode (u32 words, all < p):
 *   [0x31414B5A, n_nodes, n_constraints, n_pvs]
 *   n_nodes x [op, a, b]   ops: 0 VAR(col a, rotation b in {0,1})  1 PUB(a)
 *        2 CONST(a) 3 IS_FIRST 4 IS_LAST 5 IS_TRANSITION
 *        6 ADD(a,b) 7 SUB(a,b) 8 MUL(a,b) 9 NEG(a)
 *        10 PERM(col a, rotation b)  11 CHAL(a)  12 EXPOSED(a)   (LogUp phase leaves)
 *        13 PREP(col a, rotation b)  (preprocessed trace: fixed at keygen, committed per AIR, the
 *           commitment is observed in the preamble and opened like the main trace)
 *   n_constraints x node index (asserted zero on every row)
 *   optional: [0x50504B5A, prep_width]
 *   optional: [0x43414B5A, cached_width]  (cached main partition, OpenVM-v1 `cached_mains`: the first cached_width columns
 *             of the main trace are committed in a tree of their own -- e.g. the program ROM, whose commitment is reused
 *             across proofs -- and the rest joins the common main commitment; `commitments.main_trace` of the reference's
 *             stored proofs holds [cached..., common])
 *   optional: [0x554C4B5A, n_int, n_int x {bus, sign, count node, n_fields, field node..., group}]
 *             (count / fields = nodes that are expressions of the current row only; interactions of one group share
 *              a permutation column group: phi_g = sum of their terms; groups are numbered 0.. in order)
 * LogUp phase: challenges gamma, beta (extension); chal vector = gamma, beta^1..beta^16 as
 * 68 base coordinates; interaction j contributes phi_j = (+/-)count / (gamma + bus + 1 +
 * sum_i beta^(i+1) f_i); the permutation matrix holds phi_j (4 base columns each) and the
 * running sum of sum_j phi_j over the rows (last 4 columns); the final sum is EXPOSED and the
 * exposed sums of all AIRs must add up to zero.
 */
#include <stdlib.h>
#include <string.h>
#include "zk_oracle.h"

#define AIR_MAGIC 0x31414B5Au
#define LOGUP_MAGIC 0x554C4B5Au
#define PREP_MAGIC 0x50504B5Au
#define CACHED_MAGIC 0x43414B5Au
#define PROOF_MAGIC 0x31504B5Au
#define PROTO_TAG 0x5A4B4831u /* "1HKZ" < p */
#define GEN 31u
#define MAX_FIELDS 32
#define MAX_LOG_FINAL_POLY 8
#define N_CHAL (4 * (1 + MAX_FIELDS))

enum { OP_VAR, OP_PUB, OP_CONST, OP_FIRST, OP_LAST, OP_TRANS, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_PERM, OP_CHAL, OP_EXPOSED, OP_PREP };

typedef uint32_t ext_t[4];

static size_t bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
static void ext_set(ext_t d, const ext_t s) { memcpy(d, s, 16); }
static void ext_add(const ext_t a, const ext_t b, ext_t o) {
    for (int k = 0; k < 4; k++) o[k] = ora_add(a[k], b[k]);
}
static void ext_sub(const ext_t a, const ext_t b, ext_t o) {
    for (int k = 0; k < 4; k++) o[k] = ora_sub(a[k], b[k]);
}
static void ext_scale(const ext_t a, uint32_t s, ext_t o) {
    for (int k = 0; k < 4; k++) o[k] = ora_mul(a[k], s);
}
static void ext_from(uint32_t a, ext_t o) {
    o[0] = a;
    o[1] = o[2] = o[3] = 0;
}
static void ext_pow(const ext_t a, uint64_t e, ext_t o) {
    ext_t r = {1, 0, 0, 0}, b;
    ext_set(b, a);
    while (e) {
        if (e & 1) ora_ext_mul(r, b, r);
        ora_ext_mul(b, b, b);
        e >>= 1;
    }
    ext_set(o, r);
}

/* ---------------------------------------------------------------- program */
typedef struct {
    uint32_t bus, sign;
    uint32_t count; /* node index */
    uint32_t n_fields;
    uint32_t fields[MAX_FIELDS]; /* node indices */
    uint32_t group;              /* permutation column group */
} interaction;

typedef struct {
    uint32_t n_nodes, n_cons, n_pvs;
    const uint32_t *nodes, *cons;
    uint32_t n_int;
    interaction *ints;
    uint32_t n_groups;
    size_t perm_width; /* 4 * (n_groups + 1), or 0 */
    uint32_t max_degree; /* of the constraints, in units of the trace length */
    unsigned log_qd;     /* quotient chunks = 2^log_qd */
    size_t prep_width; /* preprocessed columns, or 0 */
    size_t cached_width; /* leading main columns committed on their own (cached main partition), or 0 */
    unsigned char *row_local; /* per node: reachable from an interaction operand (evaluated per trace row) */
} program;

static int parse_program(const uint32_t *w, size_t len, size_t width, program *p) {
    memset(p, 0, sizeof *p);
    if (len < 4 || w[0] != AIR_MAGIC) return -1;
    p->n_nodes = w[1];
    p->n_cons = w[2];
    p->n_pvs = w[3];
    size_t base_len = (size_t)4 + 3 * (size_t)p->n_nodes + p->n_cons;
    if (base_len > len) return -1;
    p->nodes = w + 4;
    p->cons = w + 4 + 3 * (size_t)p->n_nodes;
    size_t q0 = base_len;
    if (q0 + 2 <= len && w[q0] == PREP_MAGIC) {
        p->prep_width = w[q0 + 1];
        if (p->prep_width == 0 || p->prep_width > (1u << 20)) return -1;
        q0 += 2;
    }
    if (q0 + 2 <= len && w[q0] == CACHED_MAGIC) {
        p->cached_width = w[q0 + 1];
        if (p->cached_width == 0 || p->cached_width >= width) return -1; /* a common part must remain */
        q0 += 2;
    }
    if (q0 != len) {
        size_t q = q0;
        if (q + 2 > len || w[q] != LOGUP_MAGIC) return -1;
        p->n_int = w[q + 1];
        q += 2;
        if (p->n_int == 0 || p->n_int > 4096) return -1;
        p->ints = (interaction *)calloc(p->n_int, sizeof(interaction));
        for (uint32_t j = 0; j < p->n_int; j++) {
            interaction *it = &p->ints[j];
            if (q + 4 > len) return -1;
            it->bus = w[q], it->sign = w[q + 1], it->count = w[q + 2], it->n_fields = w[q + 3];
            q += 4;
            if (it->sign > 1 || it->bus >= ORA_P - 1 || it->n_fields < 1 || it->n_fields > MAX_FIELDS || q + it->n_fields > len) return -1;
            if (it->count >= p->n_nodes) return -1;
            for (uint32_t i = 0; i < it->n_fields; i++) {
                it->fields[i] = w[q++];
                if (it->fields[i] >= p->n_nodes) return -1;
            }
            if (q + 1 > len) return -1;
            it->group = w[q++];
            if (it->group != (j == 0 ? 0 : p->ints[j - 1].group) && it->group != (j == 0 ? 0 : p->ints[j - 1].group + 1)) return -1;
        }
        p->n_groups = p->ints[p->n_int - 1].group + 1;
        if (q != len) return -1;
        p->perm_width = 4 * ((size_t)p->n_groups + 1);
    }
    for (uint32_t i = 0; i < p->n_nodes; i++) {
        uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
        switch (op) {
        case OP_VAR: if (a >= width || b > 1) return -1; break;
        case OP_PUB: if (a >= p->n_pvs) return -1; break;
        case OP_CONST: if (a >= ORA_P) return -1; break;
        case OP_FIRST: case OP_LAST: case OP_TRANS: break;
        case OP_ADD: case OP_SUB: case OP_MUL: if (a >= i || b >= i) return -1; break;
        case OP_NEG: if (a >= i) return -1; break;
        case OP_PERM: if (a >= p->perm_width || b > 1) return -1; break;
        case OP_CHAL: if (a >= N_CHAL || p->n_int == 0) return -1; break;
        case OP_EXPOSED: if (a >= 4 || p->n_int == 0) return -1; break;
        case OP_PREP: if (a >= p->prep_width || b > 1) return -1; break;
        default: return -1;
        }
    }
    for (uint32_t i = 0; i < p->n_cons; i++)
        if (p->cons[i] >= p->n_nodes) return -1;
    { /* Degree of every constraint in units of the trace length (a trace cell, is_first, is_last: 1; is_transition = X - w^-1,
       * constants, public values, challenges: 0).  The quotient of an AIR of constraint degree d has degree < (d - 1) N: it is
       * split into qd = 2^ceil(log2(max(d, 2) - 1)) chunks (the rule of the reference's engine; the stored v1 proofs have
       * 1 or 4 chunks per AIR at blow-up 4). */
        uint32_t *deg = (uint32_t *)calloc(p->n_nodes + 1, sizeof(uint32_t));
        for (uint32_t i = 0; i < p->n_nodes; i++) {
            uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
            switch (op) {
            case OP_VAR: case OP_PERM: case OP_PREP: case OP_FIRST: case OP_LAST: deg[i] = 1; break;
            case OP_ADD: case OP_SUB: deg[i] = deg[a] > deg[b] ? deg[a] : deg[b]; break;
            case OP_MUL: deg[i] = deg[a] + deg[b]; if (deg[i] > 64) deg[i] = 64; break;
            case OP_NEG: deg[i] = deg[a]; break;
            default: deg[i] = 0;
            }
        }
        p->max_degree = 0;
        for (uint32_t i = 0; i < p->n_cons; i++)
            if (deg[p->cons[i]] > p->max_degree) p->max_degree = deg[p->cons[i]];
        free(deg);
        p->log_qd = 0;
        while (((uint32_t)1 << p->log_qd) + 1 < (p->max_degree < 2 ? 2 : p->max_degree)) p->log_qd++;
    }
    if (p->n_int) {
        /* operands must be expressions of the current row: mark what they reach, reject anything else */
        p->row_local = (unsigned char *)calloc(p->n_nodes, 1);
        for (uint32_t j = 0; j < p->n_int; j++) {
            p->row_local[p->ints[j].count] = 1;
            for (uint32_t i = 0; i < p->ints[j].n_fields; i++) p->row_local[p->ints[j].fields[i]] = 1;
        }
        for (uint32_t i = p->n_nodes; i-- > 0;) {
            if (!p->row_local[i]) continue;
            uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
            switch (op) {
            case OP_VAR: case OP_PREP: if (b != 0) return -1; break;
            case OP_PUB: case OP_CONST: break;
            case OP_ADD: case OP_SUB: case OP_MUL: p->row_local[a] = p->row_local[b] = 1; break;
            case OP_NEG: p->row_local[a] = 1; break;
            default: return -1;
            }
        }
    }
    return 0;
}
static void free_program(program *p) { free(p->ints), free(p->row_local); }

typedef struct {
    const uint32_t *local, *next, *pvs, *perm_local, *perm_next, *chal, *exposed;
    uint32_t is_first, is_last, is_trans;
    const uint32_t *prep_local, *prep_next;
} row_ctx;

/* base-field evaluation of all nodes on one row; vals has n_nodes slots */
static void eval_nodes_base(const program *p, const row_ctx *c, uint32_t *vals) {
    for (uint32_t i = 0; i < p->n_nodes; i++) {
        uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
        switch (op) {
        case OP_VAR: vals[i] = b ? c->next[a] : c->local[a]; break;
        case OP_PUB: vals[i] = c->pvs[a]; break;
        case OP_CONST: vals[i] = a; break;
        case OP_FIRST: vals[i] = c->is_first; break;
        case OP_LAST: vals[i] = c->is_last; break;
        case OP_TRANS: vals[i] = c->is_trans; break;
        case OP_ADD: vals[i] = ora_add(vals[a], vals[b]); break;
        case OP_SUB: vals[i] = ora_sub(vals[a], vals[b]); break;
        case OP_MUL: vals[i] = ora_mul(vals[a], vals[b]); break;
        case OP_NEG: vals[i] = ora_sub(0, vals[a]); break;
        case OP_PERM: vals[i] = b ? c->perm_next[a] : c->perm_local[a]; break;
        case OP_CHAL: vals[i] = c->chal[a]; break;
        case OP_PREP: vals[i] = b ? c->prep_next[a] : c->prep_local[a]; break;
        default: vals[i] = c->exposed[a]; break;
        }
    }
}
typedef struct {
    const ext_t *local, *next, *perm_local, *perm_next;
    const uint32_t *pvs, *chal, *exposed;
    ext_t is_first, is_last, is_trans;
    const ext_t *prep_local, *prep_next;
} zeta_ctx;
/* same over the extension (verifier, at zeta) */
static void eval_nodes_ext(const program *p, const zeta_ctx *c, ext_t *vals) {
    for (uint32_t i = 0; i < p->n_nodes; i++) {
        uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
        switch (op) {
        case OP_VAR: ext_set(vals[i], b ? c->next[a] : c->local[a]); break;
        case OP_PUB: ext_from(c->pvs[a], vals[i]); break;
        case OP_CONST: ext_from(a, vals[i]); break;
        case OP_FIRST: ext_set(vals[i], c->is_first); break;
        case OP_LAST: ext_set(vals[i], c->is_last); break;
        case OP_TRANS: ext_set(vals[i], c->is_trans); break;
        case OP_ADD: ext_add(vals[a], vals[b], vals[i]); break;
        case OP_SUB: ext_sub(vals[a], vals[b], vals[i]); break;
        case OP_MUL: ora_ext_mul(vals[a], vals[b], vals[i]); break;
        case OP_NEG: {
            ext_t z = {0, 0, 0, 0};
            ext_sub(z, vals[a], vals[i]);
        } break;
        case OP_PERM: ext_set(vals[i], b ? c->perm_next[a] : c->perm_local[a]); break;
        case OP_CHAL: ext_from(c->chal[a], vals[i]); break;
        case OP_PREP: ext_set(vals[i], b ? c->prep_next[a] : c->prep_local[a]); break;
        default: ext_from(c->exposed[a], vals[i]); break;
        }
    }
}

/* ---------------------------------------------------------------- helpers */
/* prep_commits: n_airs x 8 words, used for the AIRs whose has_prep[a] is set */
static void observe_preamble(ora_challenger *ch, const ora_params *prm, const ora_air_instance *airs,
                             size_t n_airs, const uint32_t *prep_commits, const int *has_prep) {
    uint32_t hdr[7] = {PROTO_TAG, (uint32_t)n_airs, prm->log_blowup, prm->log_final_poly_len,
                       prm->num_queries, prm->commit_pow_bits, prm->query_pow_bits};
    ora_ch_observe(ch, hdr, 7);
    for (size_t a = 0; a < n_airs; a++) {
        uint32_t dig[8];
        ora_hash_slice(airs[a].program, airs[a].program_len, dig);
        uint32_t meta[3] = {airs[a].log_height, (uint32_t)airs[a].width, (uint32_t)airs[a].n_pvs};
        ora_ch_observe(ch, meta, 3);
        ora_ch_observe(ch, dig, 8);
        if (has_prep[a]) ora_ch_observe(ch, prep_commits + 8 * a, 8);
        ora_ch_observe(ch, airs[a].pvs, airs[a].n_pvs);
    }
}

/* chal vector: gamma, beta^1 .. beta^MAX_FIELDS as base coordinates */
static void make_chal(const ext_t gamma, const ext_t beta, uint32_t *chal) {
    memcpy(chal, gamma, 16);
    ext_t cur;
    ext_set(cur, beta);
    for (int i = 1; i <= MAX_FIELDS; i++) {
        memcpy(chal + 4 * i, cur, 16);
        ora_ext_mul(cur, beta, cur);
    }
}

/* evaluate the polynomial with the given evaluations over shift*H (natural order) at ext point z */
static void eval_poly_at(const uint32_t *evals, unsigned log_n, uint32_t shift, const ext_t z, ext_t out) {
    size_t n = (size_t)1 << log_n;
    uint32_t *c = (uint32_t *)malloc(n * sizeof(uint32_t));
    memcpy(c, evals, n * sizeof(uint32_t));
    ora_dft_batch(c, log_n, 1, n, 1); /* coefficients of f(x) = p(shift*x) */
    ext_t y, acc = {0, 0, 0, 0};
    ext_scale(z, ora_inv(shift), y);
    for (size_t i = n; i-- > 0;) {
        ora_ext_mul(acc, y, acc);
        acc[0] = ora_add(acc[0], c[i]);
    }
    ext_set(out, acc);
    free(c);
}

/* selectors of the trace domain H (size 2^lh) at an extension point */
static void selectors_ext(unsigned lh, const ext_t x, ext_t is_first, ext_t is_last, ext_t is_trans, ext_t inv_zh) {
    ext_t xn, one = {1, 0, 0, 0}, zh, d, di;
    ext_pow(x, (uint64_t)1 << lh, xn);
    ext_sub(xn, one, zh);
    uint32_t winv = ora_inv(ora_two_adic_generator(lh));
    ext_sub(x, one, d);
    ora_ext_inv(d, di);
    ora_ext_mul(zh, di, is_first);
    ext_t wi = {winv, 0, 0, 0};
    ext_sub(x, wi, is_trans);
    ora_ext_inv(is_trans, di);
    ora_ext_mul(zh, di, is_last);
    ora_ext_inv(zh, inv_zh);
}

/* values of the row-local nodes (interaction operands) on one trace row */
static void eval_row_local(const program *p, const uint32_t *trace, const uint32_t *prep, const uint32_t *pvs, size_t N,
                           size_t row, uint32_t *vals) {
    for (uint32_t i = 0; i < p->n_nodes; i++) {
        if (!p->row_local[i]) continue;
        uint32_t op = p->nodes[3 * i], a = p->nodes[3 * i + 1], b = p->nodes[3 * i + 2];
        switch (op) {
        case OP_VAR: vals[i] = trace[(size_t)a * N + row]; break;
        case OP_PREP: vals[i] = prep[(size_t)a * N + row]; break;
        case OP_PUB: vals[i] = pvs[a]; break;
        case OP_CONST: vals[i] = a; break;
        case OP_ADD: vals[i] = ora_add(vals[a], vals[b]); break;
        case OP_SUB: vals[i] = ora_sub(vals[a], vals[b]); break;
        case OP_MUL: vals[i] = ora_mul(vals[a], vals[b]); break;
        default: vals[i] = ora_sub(0, vals[a]); break;
        }
    }
}

/* ---------------------------------------------------------------- prover */
typedef struct {
    uint32_t *lde;       /* column-major, height M */
    uint32_t *perm;      /* permutation trace, column-major height N (NULL if no interactions) */
    uint32_t *perm_lde;  /* its LDE, height M */
    uint32_t *qlde;      /* 2^b chunk matrices of width 4, each column-major height M, back to back */
    unsigned lh, h;      /* trace / LDE log heights */
    size_t width;
    uint32_t exposed[4];
    program prog;
    uint32_t *prep_lde;  /* preprocessed LDE (height M) and its own commitment */
    ora_tree *t_prep;
    size_t cw;           /* cached main partition: columns [0, cw) have their own tree */
    ora_tree *t_cached;
    uint32_t root_cached[8];
} air_state;

/* one committed matrix as seen by the opening / FRI code */
typedef struct {
    const uint32_t *lde;  /* column-major height 2^h */
    const uint32_t *nat;  /* natural-order evaluations over nat_shift*H (height 2^lh), for the opening at zeta */
    uint32_t nat_shift;
    unsigned lh, h;
    size_t width;
    unsigned n_pts;       /* 2: zeta and zeta*w_N, 1: zeta */
    int round;            /* 0 main, 1 perm, 2 quotient, 3 preprocessed, 4 cached main partition */
} cmat;

/* LDE + commitment of one AIR's preprocessed trace (what keygen stores in the verifying key) */
static ora_tree *commit_prep(const ora_params *prm, const ora_air_instance *air, size_t prep_width, uint32_t **lde_out,
                             uint32_t root[8]) {
    const unsigned lh = air->log_height, b = prm->log_blowup;
    const size_t N = (size_t)1 << lh, M = N << b;
    uint32_t *lde = (uint32_t *)malloc(M * prep_width * sizeof(uint32_t));
    ora_coset_lde_batch(air->prep, N, lde, M, lh, b, prep_width, GEN, 1);
    ora_matrix m = {lde, M, lh + b, prep_width};
    ora_tree *t = ora_mmcs_commit(&m, 1, root);
    *lde_out = lde;
    return t;
}

int ora_prep_commit(const ora_params *prm, const ora_air_instance *air, uint32_t root[8]) {
    program pg;
    if (parse_program(air->program, air->program_len, air->width, &pg) || !pg.prep_width || !air->prep) return -1;
    uint32_t *lde;
    ora_tree *t = commit_prep(prm, air, pg.prep_width, &lde, root);
    ora_tree_free(t);
    free(lde);
    free_program(&pg);
    return 0;
}

size_t ora_stark_prove(const ora_params *prm, const ora_air_instance *airs, size_t n_airs, uint32_t *out,
                       size_t cap) {
    const unsigned b = prm->log_blowup, nch_lde = 1u << b; /* rows of the LDE per trace row */
    const unsigned lfp = prm->log_final_poly_len; /* the fold loop stops at 2^(b+lfp) values = a polynomial of degree < 2^lfp */
    if (lfp > MAX_LOG_FINAL_POLY || n_airs == 0 || b == 0) return 0;
    air_state *st = (air_state *)calloc(n_airs, sizeof(air_state));
    unsigned hmax = 0;
    size_t n_lu = 0, n_prep = 0, n_cached = 0;
    uint32_t *prep_roots = (uint32_t *)calloc(n_airs, 32);
    int *has_prep = (int *)calloc(n_airs, sizeof(int));
    for (size_t a = 0; a < n_airs; a++) {
        if (airs[a].log_height > 27) return 0;
        if (airs[a].log_height < lfp) return 0; /* shorter than the final polynomial: its openings would never join the fold */
        if (parse_program(airs[a].program, airs[a].program_len, airs[a].width, &st[a].prog)) return 0;
        if (st[a].prog.n_pvs != airs[a].n_pvs) return 0;
        st[a].lh = airs[a].log_height;
        st[a].h = st[a].lh + b;
        st[a].width = airs[a].width;
        if (st[a].h > hmax) hmax = st[a].h;
        if (st[a].prog.n_int) n_lu++;
        st[a].cw = st[a].prog.cached_width;
        if (st[a].cw) n_cached++;
        if (st[a].prog.prep_width) {
            if (!airs[a].prep) return 0;
            st[a].t_prep = commit_prep(prm, &airs[a], st[a].prog.prep_width, &st[a].prep_lde, prep_roots + 8 * a);
            has_prep[a] = 1;
            n_prep++;
        }
    }
    ora_challenger ch;
    ora_ch_init(&ch);
    observe_preamble(&ch, prm, airs, n_airs, prep_roots, has_prep);

    /* 1. main LDE + commit */
    ora_matrix *mm = (ora_matrix *)calloc(n_airs, sizeof(ora_matrix));
    for (size_t a = 0; a < n_airs; a++) {
        size_t N = (size_t)1 << st[a].lh, M = N << b;
        st[a].lde = (uint32_t *)malloc(M * st[a].width * sizeof(uint32_t));
        ora_coset_lde_batch(airs[a].trace, N, st[a].lde, M, st[a].lh, b, st[a].width, GEN, 1);
        /* a cached partition gets its own tree; only the remaining columns join the common main commitment */
        mm[a] = (ora_matrix){st[a].lde + st[a].cw * M, M, st[a].h, st[a].width - st[a].cw};
        if (st[a].cw) {
            ora_matrix cmx = {st[a].lde, M, st[a].h, st[a].cw};
            st[a].t_cached = ora_mmcs_commit(&cmx, 1, st[a].root_cached);
        }
    }
    uint32_t root_main[8], root_perm[8], root_quot[8];
    ora_tree *t_main = ora_mmcs_commit(mm, n_airs, root_main), *t_perm = NULL;
    for (size_t a = 0; a < n_airs; a++) /* main-trace commitments in the reference's order: cached..., common */
        if (st[a].cw) ora_ch_observe(&ch, st[a].root_cached, 8);
    ora_ch_observe(&ch, root_main, 8);

    /* 1b. LogUp phase */
    uint32_t chal[N_CHAL];
    memset(chal, 0, sizeof chal);
    ora_matrix *pm = (ora_matrix *)calloc(n_airs, sizeof(ora_matrix));
    if (n_lu) {
        ext_t gamma, beta;
        ora_ch_sample_ext(&ch, gamma);
        ora_ch_sample_ext(&ch, beta);
        make_chal(gamma, beta, chal);
        size_t k = 0;
        for (size_t a = 0; a < n_airs; a++) {
            const program *pg = &st[a].prog;
            if (!pg->n_int) continue;
            const size_t N = (size_t)1 << st[a].lh, M = N << b, PW = pg->perm_width;
            st[a].perm = (uint32_t *)calloc(PW * N, sizeof(uint32_t));
            ext_t run = {0, 0, 0, 0};
            uint32_t *rv = (uint32_t *)malloc(pg->n_nodes * sizeof(uint32_t));
            for (size_t r = 0; r < N; r++) {
                ext_t rowsum = {0, 0, 0, 0};
                eval_row_local(pg, airs[a].trace, airs[a].prep, airs[a].pvs, N, r, rv);
                for (uint32_t j = 0; j < pg->n_int; j++) {
                    const interaction *it = &pg->ints[j];
                    ext_t den, inv, phi;
                    memcpy(den, chal, 16);
                    den[0] = ora_add(den[0], it->bus + 1);
                    for (uint32_t i = 0; i < it->n_fields; i++) {
                        ext_t t;
                        ext_scale(chal + 4 * (i + 1), rv[it->fields[i]], t);
                        ext_add(den, t, den);
                    }
                    ora_ext_inv(den, inv);
                    uint32_t cnt = rv[it->count];
                    if (it->sign) cnt = ora_sub(0, cnt);
                    ext_scale(inv, cnt, phi);
                    for (int q = 0; q < 4; q++) {  /* the column group accumulates the terms of its interactions */
                        uint32_t *cell = &st[a].perm[(4 * (size_t)it->group + q) * N + r];
                        *cell = ora_add(*cell, phi[q]);
                    }
                    ext_add(rowsum, phi, rowsum);
                }
                ext_add(run, rowsum, run);
                for (int q = 0; q < 4; q++) st[a].perm[(4 * (size_t)pg->n_groups + q) * N + r] = run[q];
            }
            free(rv);
            memcpy(st[a].exposed, run, 16);
            st[a].perm_lde = (uint32_t *)malloc(M * PW * sizeof(uint32_t));
            ora_coset_lde_batch(st[a].perm, N, st[a].perm_lde, M, st[a].lh, b, PW, GEN, 1);
            pm[k++] = (ora_matrix){st[a].perm_lde, M, st[a].h, PW};
        }
        t_perm = ora_mmcs_commit(pm, n_lu, root_perm);
        ora_ch_observe(&ch, root_perm, 8);
        for (size_t a = 0; a < n_airs; a++)
            if (st[a].prog.n_int) ora_ch_observe(&ch, st[a].exposed, 4);
    }
    ext_t alpha;
    ora_ch_sample_ext(&ch, alpha);

    /* 2. quotient: evaluate constraints on the LDE domain, divide by Z_H, split, LDE each chunk */
    /* AIR a has qd_a = 2^log_qd chunks (parse_program); its chunk j is quotient matrix qoff[a] + j */
    size_t *qoff = (size_t *)calloc(n_airs + 1, sizeof(size_t));
    for (size_t a = 0; a < n_airs; a++) {
        if (st[a].prog.log_qd > b) return 0; /* constraint degree needs a larger blow-up */
        qoff[a + 1] = qoff[a] + ((size_t)1 << st[a].prog.log_qd);
    }
    const size_t n_quot = qoff[n_airs];
    ora_matrix *qm = (ora_matrix *)calloc(n_quot, sizeof(ora_matrix));
    uint32_t **qnat = (uint32_t **)calloc(n_quot, sizeof(uint32_t *));
    int ok = 1;
    for (size_t a = 0; a < n_airs; a++) {
        const program *pg = &st[a].prog;
        const unsigned lh = st[a].lh, h = st[a].h;
        const size_t N = (size_t)1 << lh, M = N << b, W = st[a].width, PW = pg->perm_width, QW = pg->prep_width;
        /* the quotient domain g * H_{N qd} is the first N * qd rows of the bit-reversed LDE */
        const unsigned nch = 1u << pg->log_qd;
        const size_t MQ = N << pg->log_qd;
        ext_t *ap = (ext_t *)malloc((pg->n_cons + 1) * sizeof(ext_t));
        { /* constraint i is weighted alpha^(n_cons-1-i) (Horner order of p3's folder) */
            ext_t cur = {1, 0, 0, 0};
            for (uint32_t i = pg->n_cons; i-- > 0;) {
                ext_set(ap[i], cur);
                ora_ext_mul(cur, alpha, cur);
            }
        }
        uint32_t *q = (uint32_t *)malloc(MQ * 4 * sizeof(uint32_t)); /* [r][4] */
        const uint32_t wM = ora_two_adic_generator(h), winv = ora_inv(ora_two_adic_generator(lh));
#pragma omp parallel
        {
            uint32_t *vals = (uint32_t *)malloc(pg->n_nodes * sizeof(uint32_t));
            uint32_t *loc = (uint32_t *)malloc((W + 1) * sizeof(uint32_t)), *nxt = (uint32_t *)malloc((W + 1) * sizeof(uint32_t));
            uint32_t *ploc = (uint32_t *)malloc((PW + 1) * sizeof(uint32_t)), *pnxt = (uint32_t *)malloc((PW + 1) * sizeof(uint32_t));
            uint32_t *qloc = (uint32_t *)malloc((QW + 1) * sizeof(uint32_t)), *qnxt = (uint32_t *)malloc((QW + 1) * sizeof(uint32_t));
#pragma omp for schedule(static)
            for (size_t r = 0; r < MQ; r++) {
                size_t i = bitrev(r, h), rn = bitrev((i + nch_lde) & (M - 1), h);
                uint32_t x = ora_mul(GEN, ora_pow(wM, i));
                for (size_t c = 0; c < W; c++) {
                    loc[c] = st[a].lde[c * M + r];
                    nxt[c] = st[a].lde[c * M + rn];
                }
                for (size_t c = 0; c < PW; c++) {
                    ploc[c] = st[a].perm_lde[c * M + r];
                    pnxt[c] = st[a].perm_lde[c * M + rn];
                }
                for (size_t c = 0; c < QW; c++) {
                    qloc[c] = st[a].prep_lde[c * M + r];
                    qnxt[c] = st[a].prep_lde[c * M + rn];
                }
                uint32_t zh = ora_sub(ora_pow(x, N), 1);
                row_ctx rc = {loc, nxt, airs[a].pvs, ploc, pnxt, chal, st[a].exposed, 0, 0, 0, qloc, qnxt};
                rc.is_first = ora_mul(zh, ora_inv(ora_sub(x, 1)));
                rc.is_trans = ora_sub(x, winv);
                rc.is_last = ora_mul(zh, ora_inv(rc.is_trans));
                eval_nodes_base(pg, &rc, vals);
                ext_t acc = {0, 0, 0, 0}, t;
                for (uint32_t k = 0; k < pg->n_cons; k++) {
                    ext_scale(ap[k], vals[pg->cons[k]], t);
                    ext_add(acc, t, acc);
                }
                ext_scale(acc, ora_inv(zh), q + 4 * r);
            }
            free(vals), free(loc), free(nxt), free(ploc), free(pnxt), free(qloc), free(qnxt);
        }
        /* chunk j = rows [jN,(j+1)N): evaluations over s_j*H (bit-reversed), s_j = g * wM^bitrev_b(j) */
        st[a].qlde = (uint32_t *)malloc((size_t)nch * 4 * M * sizeof(uint32_t));
        for (unsigned j = 0; j < nch; j++) {
            uint32_t *nat = (uint32_t *)malloc(4 * N * sizeof(uint32_t));
#pragma omp parallel for schedule(static)
            for (size_t m = 0; m < N; m++)
                for (int k = 0; k < 4; k++) nat[k * N + m] = q[4 * (j * N + bitrev(m, lh)) + k];
            uint32_t sj = ora_mul(GEN, ora_pow(wM, bitrev(j, b)));
            uint32_t *dst = st[a].qlde + (size_t)j * 4 * M;
            ora_coset_lde_batch(nat, N, dst, M, lh, b, 4, ora_mul(GEN, ora_inv(sj)), 1);
            qm[qoff[a] + j] = (ora_matrix){dst, M, h, 4};
            qnat[qoff[a] + j] = nat;
        }
        free(q), free(ap);
    }
    ora_tree *t_quot = ora_mmcs_commit(qm, n_quot, root_quot);
    ora_ch_observe(&ch, root_quot, 8);
    ext_t zeta;
    ora_ch_sample_ext(&ch, zeta);

    /* committed matrices in opening order: main (all AIRs), preprocessed (AIRs that have one), perm (AIRs with
     * interactions), quotient chunks */
    size_t n_cm = n_airs + n_cached + n_prep + n_lu + n_quot;
    cmat *cm = (cmat *)calloc(n_cm, sizeof(cmat));
    {
        size_t k = 0;
        for (size_t a = 0; a < n_airs; a++)
            cm[k++] = (cmat){st[a].lde + st[a].cw * ((size_t)1 << st[a].h), airs[a].trace + st[a].cw * ((size_t)1 << st[a].lh), 1,
                             st[a].lh, st[a].h, st[a].width - st[a].cw, 2, 0};
        for (size_t a = 0; a < n_airs; a++)
            if (st[a].cw) cm[k++] = (cmat){st[a].lde, airs[a].trace, 1, st[a].lh, st[a].h, st[a].cw, 2, 4};
        for (size_t a = 0; a < n_airs; a++)
            if (has_prep[a]) cm[k++] = (cmat){st[a].prep_lde, airs[a].prep, 1, st[a].lh, st[a].h, st[a].prog.prep_width, 2, 3};
        for (size_t a = 0; a < n_airs; a++)
            if (st[a].prog.n_int) cm[k++] = (cmat){st[a].perm_lde, st[a].perm, 1, st[a].lh, st[a].h, st[a].prog.perm_width, 2, 1};
        for (size_t a = 0; a < n_airs; a++) {
            const uint32_t wM = ora_two_adic_generator(st[a].h);
            for (unsigned j = 0; j < (1u << st[a].prog.log_qd); j++)
                cm[k++] = (cmat){st[a].qlde + (size_t)j * 4 * ((size_t)1 << st[a].h), qnat[qoff[a] + j],
                                 ora_mul(GEN, ora_pow(wM, bitrev(j, b))), st[a].lh, st[a].h, 4, 1, 2};
        }
    }

    /* 3. openings */
    size_t n_open = 0;
    for (size_t m = 0; m < n_cm; m++) n_open += cm[m].width * cm[m].n_pts;
    ext_t *opened = (ext_t *)malloc(n_open * sizeof(ext_t));
    {
        size_t oi = 0;
        for (size_t m = 0; m < n_cm; m++) {
            const size_t N = (size_t)1 << cm[m].lh, W = cm[m].width;
            for (unsigned pt = 0; pt < cm[m].n_pts; pt++) {
                ext_t z;
                if (pt == 0) ext_set(z, zeta);
                else ext_scale(zeta, ora_two_adic_generator(cm[m].lh), z);
#pragma omp parallel for schedule(dynamic)
                for (size_t c = 0; c < W; c++) eval_poly_at(cm[m].nat + c * N, cm[m].lh, cm[m].nat_shift, z, opened[oi + c]);
                oi += W;
            }
        }
    }
    ora_ch_observe(&ch, (const uint32_t *)opened, 4 * n_open);
    ext_t alpha_f;
    ora_ch_sample_ext(&ch, alpha_f);

    /* 4. reduced openings per LDE log-height */
    ext_t **ro = (ext_t **)calloc(hmax + 1, sizeof(ext_t *));
    size_t *num_reduced = (size_t *)calloc(hmax + 1, sizeof(size_t));
    {
        size_t oi = 0;
        for (size_t m = 0; m < n_cm; m++) {
            const unsigned h = cm[m].h;
            const size_t M = (size_t)1 << h, W = cm[m].width;
            const uint32_t *mat = cm[m].lde;
            if (!ro[h]) ro[h] = (ext_t *)calloc(M, sizeof(ext_t));
            ext_t *apow = (ext_t *)malloc(W * sizeof(ext_t));
            {
                ext_t cur = {1, 0, 0, 0};
                for (size_t k = 0; k < W; k++) {
                    ext_set(apow[k], cur);
                    ora_ext_mul(cur, alpha_f, cur);
                }
            }
            ext_t *rrow = (ext_t *)malloc(M * sizeof(ext_t));
#pragma omp parallel for schedule(static)
            for (size_t r = 0; r < M; r++) {
                ext_t acc = {0, 0, 0, 0}, t;
                for (size_t k = 0; k < W; k++) {
                    ext_scale(apow[k], mat[k * M + r], t);
                    ext_add(acc, t, acc);
                }
                ext_set(rrow[r], acc);
            }
            const uint32_t wM = ora_two_adic_generator(h);
            for (unsigned pt = 0; pt < cm[m].n_pts; pt++) {
                ext_t z, ry = {0, 0, 0, 0}, off, t;
                if (pt == 0) ext_set(z, zeta);
                else ext_scale(zeta, ora_two_adic_generator(cm[m].lh), z);
                for (size_t k = 0; k < W; k++) {
                    ora_ext_mul(apow[k], opened[oi + k], t);
                    ext_add(ry, t, ry);
                }
                ext_pow(alpha_f, num_reduced[h], off);
#pragma omp parallel for schedule(static)
                for (size_t r = 0; r < M; r++) {
                    uint32_t x = ora_mul(GEN, ora_pow(wM, bitrev(r, h)));
                    ext_t d, di, num, u;
                    ext_set(d, z);
                    d[0] = ora_sub(d[0], x);
                    ora_ext_inv(d, di);
                    ext_sub(ry, rrow[r], num);
                    ora_ext_mul(num, di, u);
                    ora_ext_mul(u, off, u);
                    ext_add(ro[h][r], u, ro[h][r]);
                }
                num_reduced[h] += W;
                oi += W;
            }
            free(apow), free(rrow);
        }
    }

    /* 5. FRI commit phase */
    const unsigned n_layers = hmax - b - lfp;
    ora_tree **ftrees = (ora_tree **)calloc(n_layers, sizeof(ora_tree *));
    uint32_t **flayers = (uint32_t **)calloc(n_layers + 1, sizeof(uint32_t *)); /* [len][4] each */
    uint32_t **fleaves = (uint32_t **)calloc(n_layers, sizeof(uint32_t *));      /* column-major 8 x len/2 */
    uint32_t(*froots)[8] = (uint32_t(*)[8])calloc(n_layers, 32);
    uint32_t *fpow = (uint32_t *)calloc(n_layers + 1, sizeof(uint32_t));
    ext_t *betas = (ext_t *)calloc(n_layers, sizeof(ext_t));
    flayers[0] = (uint32_t *)ro[hmax];
    for (unsigned l = 0; l < n_layers; l++) {
        const unsigned log_len = hmax - l;
        const size_t half = (size_t)1 << (log_len - 1);
        fleaves[l] = (uint32_t *)malloc(8 * half * sizeof(uint32_t));
#pragma omp parallel for schedule(static) if (half > 4096)
        for (size_t i = 0; i < half; i++)
            for (int k = 0; k < 8; k++) fleaves[l][k * half + i] = flayers[l][8 * i + k];
        ora_matrix lm = {fleaves[l], half, log_len - 1, 8};
        ftrees[l] = ora_mmcs_commit(&lm, 1, froots[l]);
        ora_ch_observe(&ch, froots[l], 8);
        fpow[l] = ora_ch_grind(&ch, prm->commit_pow_bits);
        ora_ch_sample_ext(&ch, betas[l]);
        flayers[l + 1] = (uint32_t *)malloc(4 * half * sizeof(uint32_t));
        ora_fri_fold(flayers[l], flayers[l + 1], log_len - 1, betas[l]);
        if (ro[log_len - 1]) {
            ext_t b2;
            ora_ext_mul(betas[l], betas[l], b2);
#pragma omp parallel for schedule(static) if (half > 4096)
            for (size_t i = 0; i < half; i++) {
                ext_t t;
                ora_ext_mul(b2, ro[log_len - 1][i], t);
                ext_add(flayers[l + 1] + 4 * i, t, flayers[l + 1] + 4 * i);
            }
        }
    }
    /* final polynomial: the last layer holds 2^(b+lfp) evaluations (bit-reversed) over the subgroup of that size of a
     * polynomial of degree < 2^lfp; its 2^lfp coefficients are sent.  lfp = 0: a constant (all values equal). */
    const size_t n_fin = (size_t)1 << lfp, n_last = (size_t)1 << (b + lfp);
    uint32_t *fin = (uint32_t *)calloc(4 * n_fin, sizeof(uint32_t));
    {
        const uint32_t *last = flayers[n_layers];
        if (lfp == 0) {
            for (size_t i = 1; i < n_last; i++)
                if (memcmp(last, last + 4 * i, 16)) ok = 0;
            memcpy(fin, last, 16);
        } else {
            uint32_t *mat = (uint32_t *)malloc(4 * n_last * sizeof(uint32_t)); /* 4 coordinate columns, natural order */
            for (size_t i = 0; i < n_last; i++)
                for (int q = 0; q < 4; q++) mat[(size_t)q * n_last + i] = last[4 * bitrev(i, b + lfp) + q];
            ora_dft_batch(mat, b + lfp, 4, n_last, 1);
            for (size_t j = 0; j < n_last; j++)
                for (int q = 0; q < 4; q++) {
                    if (j < n_fin) fin[4 * j + q] = mat[(size_t)q * n_last + j];
                    else if (mat[(size_t)q * n_last + j]) ok = 0; /* not low-degree: unsatisfied witness */
                }
            free(mat);
        }
    }
    ora_ch_observe(&ch, fin, 4 * n_fin);
    uint32_t qpow = ora_ch_grind(&ch, prm->query_pow_bits);

    /* 6. assemble the proof */
    size_t w = 0;
#define PUT(ptr, n)                                     \
    do {                                                \
        if (w + (n) > cap) { ok = 0; goto done; }       \
        memcpy(out + w, (ptr), (size_t)(n) * 4);        \
        w += (n);                                       \
    } while (0)
    {
        uint32_t hdr[4] = {PROOF_MAGIC + (n_lu ? 1u : 0u) + (n_prep ? 2u : 0u) + (n_cached ? 4u : 0u), (uint32_t)n_airs, hmax, n_layers};
        PUT(hdr, 4);
        PUT(root_main, 8);
        for (size_t a = 0; a < n_airs; a++)
            if (st[a].cw) PUT(st[a].root_cached, 8);
        if (n_lu) {
            PUT(root_perm, 8);
            for (size_t a = 0; a < n_airs; a++)
                if (st[a].prog.n_int) PUT(st[a].exposed, 4);
        }
        PUT(root_quot, 8);
        PUT(opened, 4 * n_open);
        for (unsigned l = 0; l < n_layers; l++) {
            PUT(froots[l], 8);
            PUT(&fpow[l], 1);
        }
        PUT(fin, 4 * n_fin);
        PUT(&qpow, 1);
        size_t tmp_words = 8 * (hmax + 1) + 16;
        for (size_t m = 0; m < n_cm; m++) tmp_words += cm[m].width;
        uint32_t *tmp = (uint32_t *)malloc(tmp_words * sizeof(uint32_t));
        for (unsigned qn = 0; qn < prm->num_queries; qn++) {
            size_t idx = ora_ch_sample_bits(&ch, hmax);
            size_t n1 = ora_mmcs_open(t_main, idx >> (hmax - ora_tree_log_height(t_main)), tmp);
            PUT(tmp, n1);
            for (size_t a = 0; a < n_airs; a++)
                if (st[a].cw) {
                    n1 = ora_mmcs_open(st[a].t_cached, idx >> (hmax - st[a].h), tmp);
                    PUT(tmp, n1);
                }
            for (size_t a = 0; a < n_airs; a++)
                if (has_prep[a]) {
                    n1 = ora_mmcs_open(st[a].t_prep, idx >> (hmax - st[a].h), tmp);
                    PUT(tmp, n1);
                }
            if (n_lu) {
                n1 = ora_mmcs_open(t_perm, idx >> (hmax - ora_tree_log_height(t_perm)), tmp);
                PUT(tmp, n1);
            }
            n1 = ora_mmcs_open(t_quot, idx >> (hmax - ora_tree_log_height(t_quot)), tmp);
            PUT(tmp, n1);
            for (unsigned l = 0; l < n_layers; l++) {
                size_t il = idx >> l;
                PUT(flayers[l] + 4 * (il ^ 1), 4);
                n1 = ora_mmcs_open(ftrees[l], il >> 1, tmp);
                PUT(tmp + 8, n1 - 8); /* skip the opened row (the pair), keep the path */
            }
        }
        free(tmp);
    }
done:
    for (size_t a = 0; a < n_airs; a++) {
        free(st[a].lde), free(st[a].qlde), free(st[a].perm), free(st[a].perm_lde), free(st[a].prep_lde);
        if (st[a].t_prep) ora_tree_free(st[a].t_prep);
        if (st[a].t_cached) ora_tree_free(st[a].t_cached);
        free_program(&st[a].prog);
    }
    for (size_t k = 0; k < n_quot; k++) free(qnat[k]);
    free(qoff);
    for (unsigned l = 0; l < n_layers; l++) {
        ora_tree_free(ftrees[l]);
        free(fleaves[l]);
        free(flayers[l + 1]);
    }
    for (unsigned h = 0; h <= hmax; h++) free(ro[h]);
    ora_tree_free(t_main), ora_tree_free(t_quot);
    if (t_perm) ora_tree_free(t_perm);
    free(ftrees), free(flayers), free(fleaves), free(froots), free(fpow), free(betas);
    free(ro), free(num_reduced), free(opened), free(mm), free(pm), free(qm), free(qnat), free(cm), free(st);
    free(prep_roots), free(has_prep), free(fin);
    return ok ? w : 0;
}

/* K5 on its own (the checker of zkhip_constraint_eval): quotient values of one AIR over its LDE domain.  lde: `width` columns of
 * 2^(lh+b) canonical values in the committed layout (bit-reversed rows, coset shift GEN); q: 4 columns (extension coordinates) of
 * 2^(lh+b), same row order.  AIRs with interactions / preprocessed traces are not handled here.  Returns 0, or -1 on a bad program. */
int ora_constraint_eval(const uint32_t *prog_words, size_t prog_len, unsigned lh, unsigned b, size_t width, const uint32_t *lde,
                        const uint32_t *pvs, size_t n_pvs, const uint32_t alpha[4], uint32_t *q) {
    program pgm;
    if (parse_program(prog_words, prog_len, width, &pgm) || pgm.n_int || pgm.prep_width || pgm.n_pvs != n_pvs) return -1;
    const program *pg = &pgm;
    const unsigned h = lh + b, nch = 1u << b;
    const size_t N = (size_t)1 << lh, M = N << b, W = width;
    ext_t *ap = (ext_t *)malloc((pg->n_cons + 1) * sizeof(ext_t));
    {
        ext_t cur = {1, 0, 0, 0};
        for (uint32_t i = pg->n_cons; i-- > 0;) {
            ext_set(ap[i], cur);
            ora_ext_mul(cur, alpha, cur);
        }
    }
    const uint32_t wM = ora_two_adic_generator(h), winv = ora_inv(ora_two_adic_generator(lh));
#pragma omp parallel
    {
        uint32_t *vals = (uint32_t *)malloc((pg->n_nodes + 1) * sizeof(uint32_t));
        uint32_t *loc = (uint32_t *)malloc((W + 1) * sizeof(uint32_t)), *nxt = (uint32_t *)malloc((W + 1) * sizeof(uint32_t));
#pragma omp for schedule(static)
        for (size_t r = 0; r < M; r++) {
            size_t i = bitrev(r, h), rn = bitrev((i + nch) & (M - 1), h);
            uint32_t x = ora_mul(GEN, ora_pow(wM, i));
            for (size_t c = 0; c < W; c++) loc[c] = lde[c * M + r], nxt[c] = lde[c * M + rn];
            uint32_t zh = ora_sub(ora_pow(x, N), 1);
            row_ctx rc = {loc, nxt, pvs, NULL, NULL, NULL, NULL, 0, 0, 0, NULL, NULL};
            rc.is_first = ora_mul(zh, ora_inv(ora_sub(x, 1)));
            rc.is_trans = ora_sub(x, winv);
            rc.is_last = ora_mul(zh, ora_inv(rc.is_trans));
            eval_nodes_base(pg, &rc, vals);
            ext_t acc = {0, 0, 0, 0}, t, out;
            for (uint32_t k = 0; k < pg->n_cons; k++) {
                ext_scale(ap[k], vals[pg->cons[k]], t);
                ext_add(acc, t, acc);
            }
            ext_scale(acc, ora_inv(zh), out);
            for (int k = 0; k < 4; k++) q[(size_t)k * M + r] = out[k];
        }
        free(vals), free(loc), free(nxt);
    }
    free(ap);
    free_program(&pgm);
    return 0;
}

/* ---------------------------------------------------------------- verifier */
typedef struct {
    unsigned lh, h;
    size_t width;
    unsigned n_pts;
} vmat;

/* The verifier's bus check on its own (OpenVM-v1 LogUp: every AIR with interactions exposes the cumulative sum of its
 * permutation column; the buses balance iff the sums of all AIRs add up to zero in the quartic extension).  Pinned to the reference:
 * the `exposed_values_after_challenge` of its eight stored proofs sum to zero (tests/test_ref_vectors_cpu.py). */
int ora_logup_exposed_check(const uint32_t *exposed, size_t n) {
    ext_t tot = {0, 0, 0, 0};
    for (size_t k = 0; k < n; k++) {
        for (int q = 0; q < 4; q++)
            if (exposed[4 * k + q] >= ORA_P) return -1;
        ext_add(tot, exposed + 4 * k, tot);
    }
    return (tot[0] | tot[1] | tot[2] | tot[3]) ? -9 : 0;
}

int ora_stark_verify(const ora_params *prm, const ora_air_instance *airs, size_t n_airs, const uint32_t *proof,
                     size_t n_words) {
    const unsigned b = prm->log_blowup;
    const unsigned lfp = prm->log_final_poly_len;
    if (lfp > MAX_LOG_FINAL_POLY || n_airs == 0 || b == 0) return -1;
    program *pg = (program *)calloc(n_airs, sizeof(program));
    unsigned hmax = 0;
    size_t n_lu = 0, n_prep = 0, n_cached = 0;
    uint32_t *prep_roots = (uint32_t *)calloc(n_airs, 32);
    int *has_prep = (int *)calloc(n_airs, sizeof(int));
    for (size_t a = 0; a < n_airs; a++) {
        if (parse_program(airs[a].program, airs[a].program_len, airs[a].width, &pg[a])) return -2;
        if (pg[a].cached_width) n_cached++;
        if (pg[a].n_pvs != airs[a].n_pvs || airs[a].log_height > 27 || airs[a].log_height < lfp) return -2;
        if (pg[a].log_qd > b) return -2; /* constraint degree above 2^log_blowup + 1 */
        if (airs[a].log_height + b > hmax) hmax = airs[a].log_height + b;
        if (pg[a].n_int) n_lu++;
        if (pg[a].prep_width) {
            /* the verifying key holds the commitment; tests may pass the table itself instead */
            if (airs[a].prep_commit) memcpy(prep_roots + 8 * a, airs[a].prep_commit, 32);
            else if (!airs[a].prep || ora_prep_commit(prm, &airs[a], prep_roots + 8 * a)) return -2;
            has_prep[a] = 1;
            n_prep++;
        }
    }
    /* committed matrices in opening order */
    size_t *qoff = (size_t *)calloc(n_airs + 1, sizeof(size_t)); /* quotient chunks per AIR: 2^log_qd (parse_program) */
    for (size_t a = 0; a < n_airs; a++) qoff[a + 1] = qoff[a] + ((size_t)1 << pg[a].log_qd);
    const size_t n_quot = qoff[n_airs];
    size_t n_cm = n_airs + n_cached + n_prep + n_lu + n_quot, n_open = 0;
    vmat *cm = (vmat *)calloc(n_cm, sizeof(vmat));
    size_t cm_cached0 = n_airs, cm_prep0 = n_airs + n_cached, cm_perm0 = cm_prep0 + n_prep, cm_quot0 = cm_perm0 + n_lu;
    {
        size_t k = 0;
        for (size_t a = 0; a < n_airs; a++)
            cm[k++] = (vmat){airs[a].log_height, airs[a].log_height + b, airs[a].width - pg[a].cached_width, 2};
        for (size_t a = 0; a < n_airs; a++)
            if (pg[a].cached_width) cm[k++] = (vmat){airs[a].log_height, airs[a].log_height + b, pg[a].cached_width, 2};
        for (size_t a = 0; a < n_airs; a++)
            if (has_prep[a]) cm[k++] = (vmat){airs[a].log_height, airs[a].log_height + b, pg[a].prep_width, 2};
        for (size_t a = 0; a < n_airs; a++)
            if (pg[a].n_int) cm[k++] = (vmat){airs[a].log_height, airs[a].log_height + b, pg[a].perm_width, 2};
        for (size_t a = 0; a < n_airs; a++)
            for (unsigned j = 0; j < (1u << pg[a].log_qd); j++) cm[k++] = (vmat){airs[a].log_height, airs[a].log_height + b, 4, 1};
        for (size_t m = 0; m < n_cm; m++) n_open += cm[m].width * cm[m].n_pts;
    }
    const unsigned n_layers = hmax - b - lfp;
    const size_t n_fin = (size_t)1 << lfp;
    size_t r = 0;
#define NEED(n)                         \
    do {                                \
        if (r + (n) > n_words) return -3; \
    } while (0)
    for (size_t i = 0; i < n_words; i++)
        if (proof[i] >= ORA_P) return -3; /* all words canonical (the magic is < p too) */
    NEED(4);
    if (proof[0] != PROOF_MAGIC + (n_lu ? 1u : 0u) + (n_prep ? 2u : 0u) + (n_cached ? 4u : 0u) || proof[1] != n_airs || proof[2] != hmax || proof[3] != n_layers) return -3;
    r = 4;
    NEED(8 + 8 * n_cached);
    const uint32_t *root_main = proof + r, *root_perm = NULL, *exposed_all = NULL;
    r += 8;
    const uint32_t *roots_cached = proof + r; /* one per AIR with a cached partition, AIR order */
    r += 8 * n_cached;
    if (n_lu) {
        NEED(8 + 4 * n_lu);
        root_perm = proof + r;
        exposed_all = proof + r + 8;
        r += 8 + 4 * n_lu;
    }
    NEED(8);
    const uint32_t *root_quot = proof + r;
    r += 8;
    NEED(4 * n_open);
    const ext_t *opened = (const ext_t *)(proof + r);
    r += 4 * n_open;
    NEED(9 * (size_t)n_layers + 4 * n_fin + 1);
    const uint32_t *fri_hdr = proof + r;
    r += 9 * (size_t)n_layers;
    const uint32_t *fin = proof + r; /* 2^lfp coefficients */
    r += 4 * n_fin;
    const uint32_t qpow = proof[r++];

    ora_challenger ch;
    ora_ch_init(&ch);
    observe_preamble(&ch, prm, airs, n_airs, prep_roots, has_prep);
    ora_ch_observe(&ch, roots_cached, 8 * n_cached);
    ora_ch_observe(&ch, root_main, 8);
    uint32_t chal[N_CHAL];
    memset(chal, 0, sizeof chal);
    if (n_lu) {
        ext_t gamma, beta;
        ora_ch_sample_ext(&ch, gamma);
        ora_ch_sample_ext(&ch, beta);
        make_chal(gamma, beta, chal);
        ora_ch_observe(&ch, root_perm, 8);
        ora_ch_observe(&ch, exposed_all, 4 * n_lu);
        /* bus balance: the exposed cumulative sums of all AIRs add up to zero */
        if (ora_logup_exposed_check(exposed_all, n_lu) != 0) return -9;
    }
    ext_t alpha, zeta, alpha_f;
    ora_ch_sample_ext(&ch, alpha);
    ora_ch_observe(&ch, root_quot, 8);
    ora_ch_sample_ext(&ch, zeta);
    ora_ch_observe(&ch, (const uint32_t *)opened, 4 * n_open);
    ora_ch_sample_ext(&ch, alpha_f);

    /* offsets of each committed matrix inside `opened` */
    size_t *open_off = (size_t *)malloc(n_cm * sizeof(size_t));
    {
        size_t o = 0;
        for (size_t m = 0; m < n_cm; m++) {
            open_off[m] = o;
            o += cm[m].width * cm[m].n_pts;
        }
    }
    /* constraint check at zeta for every AIR */
    {
        size_t k_lu = 0, k_prep = 0, k_cached = 0;
        for (size_t a = 0; a < n_airs; a++) {
            const unsigned lh = airs[a].log_height, h = lh + b;
            const size_t W = airs[a].width, CW = pg[a].cached_width;
            zeta_ctx zc;
            memset(&zc, 0, sizeof zc);
            ext_t inv_zh;
            selectors_ext(lh, zeta, zc.is_first, zc.is_last, zc.is_trans, inv_zh);
            /* the AIR's main row = cached partition columns, then the common columns */
            ext_t *mrow = (ext_t *)malloc(2 * W * sizeof(ext_t));
            if (CW) {
                const ext_t *co = opened + open_off[cm_cached0 + k_cached++];
                memcpy(mrow, co, CW * sizeof(ext_t));
                memcpy(mrow + W, co + CW, CW * sizeof(ext_t));
            }
            memcpy(mrow + CW, opened + open_off[a], (W - CW) * sizeof(ext_t));
            memcpy(mrow + W + CW, opened + open_off[a] + (W - CW), (W - CW) * sizeof(ext_t));
            zc.local = (const ext_t *)mrow;
            zc.next = (const ext_t *)(mrow + W);
            zc.pvs = airs[a].pvs;
            zc.chal = chal;
            if (has_prep[a]) {
                zc.prep_local = opened + open_off[cm_prep0 + k_prep];
                zc.prep_next = zc.prep_local + pg[a].prep_width;
                k_prep++;
            }
            if (pg[a].n_int) {
                zc.perm_local = opened + open_off[cm_perm0 + k_lu];
                zc.perm_next = zc.perm_local + pg[a].perm_width;
                zc.exposed = exposed_all + 4 * k_lu;
                k_lu++;
            }
            ext_t *vals = (ext_t *)malloc(pg[a].n_nodes * sizeof(ext_t));
            eval_nodes_ext(&pg[a], &zc, vals);
            free(mrow);
            ext_t acc = {0, 0, 0, 0};
            for (uint32_t k = 0; k < pg[a].n_cons; k++) {
                ora_ext_mul(acc, alpha, acc);
                ext_add(acc, vals[pg[a].cons[k]], acc);
            }
            free(vals);
            ext_t lhs;
            ora_ext_mul(acc, inv_zh, lhs);
            /* quotient(zeta) = sum_j zps_j * sum_k x^k * chunk_{j,k}(zeta) */
            const uint32_t wM = ora_two_adic_generator(h);
            ext_t rhs = {0, 0, 0, 0};
            const unsigned nch = 1u << pg[a].log_qd;
            for (unsigned j = 0; j < nch; j++) {
                uint32_t sj = ora_mul(GEN, ora_pow(wM, bitrev(j, b)));
                ext_t zps = {1, 0, 0, 0};
                for (unsigned k = 0; k < nch; k++) {
                    if (k == j) continue;
                    uint32_t sk = ora_mul(GEN, ora_pow(wM, bitrev(k, b)));
                    /* Z_{D_k}(x) = (x/s_k)^N - 1 */
                    ext_t t, one = {1, 0, 0, 0}, num;
                    ext_scale(zeta, ora_inv(sk), t);
                    ext_pow(t, (uint64_t)1 << lh, t);
                    ext_sub(t, one, num);
                    uint32_t den = ora_sub(ora_pow(ora_mul(sj, ora_inv(sk)), (uint64_t)1 << lh), 1);
                    ext_scale(num, ora_inv(den), num);
                    ora_ext_mul(zps, num, zps);
                }
                const ext_t *chunk = opened + open_off[cm_quot0 + qoff[a] + j];
                ext_t v = {0, 0, 0, 0};
                for (int k = 0; k < 4; k++) {
                    /* basis element x^k times an extension value */
                    ext_t e = {0, 0, 0, 0}, t;
                    e[k] = 1;
                    ora_ext_mul(e, chunk[k], t);
                    ext_add(v, t, v);
                }
                ora_ext_mul(v, zps, v);
                ext_add(rhs, v, rhs);
            }
            if (memcmp(lhs, rhs, 16)) return -4;
        }
    }

    /* FRI transcript */
    ext_t *betas = (ext_t *)calloc(n_layers ? n_layers : 1, sizeof(ext_t));
    for (unsigned l = 0; l < n_layers; l++) {
        ora_ch_observe(&ch, fri_hdr + 9 * l, 8);
        if (!ora_ch_check_witness(&ch, prm->commit_pow_bits, fri_hdr[9 * l + 8])) return -5;
        ora_ch_sample_ext(&ch, betas[l]);
    }
    ora_ch_observe(&ch, fin, 4 * n_fin);
    if (!ora_ch_check_witness(&ch, prm->query_pow_bits, qpow)) return -5;

    /* shapes of the input batches: main, one per preprocessed trace, perm, quotient */
    const size_t n_batches = 2 + n_cached + n_prep + (n_lu ? 1 : 0);
    size_t *batch_first = (size_t *)malloc(n_batches * sizeof(size_t)), *batch_n = (size_t *)malloc(n_batches * sizeof(size_t));
    const uint32_t **batch_root = (const uint32_t **)malloc(n_batches * sizeof(uint32_t *));
    const uint32_t **rows_of = (const uint32_t **)malloc(n_batches * sizeof(uint32_t *));
    {
        size_t bt = 0, k = 0;
        batch_first[bt] = 0, batch_n[bt] = n_airs, batch_root[bt] = root_main, bt++;
        for (size_t a = 0, kc = 0; a < n_airs; a++)
            if (pg[a].cached_width) batch_first[bt] = cm_cached0 + kc, batch_n[bt] = 1, batch_root[bt] = roots_cached + 8 * kc, kc++, bt++;
        for (size_t a = 0; a < n_airs; a++)
            if (has_prep[a]) batch_first[bt] = cm_prep0 + k++, batch_n[bt] = 1, batch_root[bt] = prep_roots + 8 * a, bt++;
        if (n_lu) batch_first[bt] = cm_perm0, batch_n[bt] = n_lu, batch_root[bt] = root_perm, bt++;
        batch_first[bt] = cm_quot0, batch_n[bt] = n_quot, batch_root[bt] = root_quot, bt++;
    }
    unsigned *lhs_all = (unsigned *)malloc(n_cm * sizeof(unsigned));
    size_t *ws_all = (size_t *)malloc(n_cm * sizeof(size_t));
    for (size_t m = 0; m < n_cm; m++) lhs_all[m] = cm[m].h, ws_all[m] = cm[m].width;
    ext_t *roq = (ext_t *)calloc(hmax + 1, sizeof(ext_t));
    int *has = (int *)calloc(hmax + 1, sizeof(int));
    size_t *num_reduced = (size_t *)calloc(hmax + 1, sizeof(size_t));
    int rc = 0;
    for (unsigned qn = 0; qn < prm->num_queries && rc == 0; qn++) {
        size_t idx = ora_ch_sample_bits(&ch, hmax);
        for (size_t bt = 0; bt < n_batches && rc == 0; bt++) {
            size_t tw = 0;
            unsigned bh = 0;
            for (size_t m = 0; m < batch_n[bt]; m++) {
                tw += ws_all[batch_first[bt] + m];
                if (lhs_all[batch_first[bt] + m] > bh) bh = lhs_all[batch_first[bt] + m];
            }
            size_t n_op = tw + 8 * (size_t)bh;
            if (r + n_op > n_words) { rc = -3; break; }
            rows_of[bt] = proof + r;
            r += n_op;
            if (!ora_mmcs_verify(batch_root[bt], lhs_all + batch_first[bt], ws_all + batch_first[bt], batch_n[bt],
                                 idx >> (hmax - bh), rows_of[bt]))
                rc = -6;
        }
        if (rc) break;
        /* reduced openings at this query for every height */
        memset(roq, 0, (hmax + 1) * sizeof(ext_t));
        memset(has, 0, (hmax + 1) * sizeof(int));
        memset(num_reduced, 0, (hmax + 1) * sizeof(size_t));
        size_t oi = 0;
        for (size_t bt = 0; bt < n_batches; bt++) {
            const uint32_t *rows = rows_of[bt];
            for (size_t mi = 0; mi < batch_n[bt]; mi++) {
                const vmat *M = &cm[batch_first[bt] + mi];
                const unsigned h = M->h;
                const size_t W = M->width;
                size_t ih = idx >> (hmax - h);
                uint32_t x = ora_mul(GEN, ora_pow(ora_two_adic_generator(h), bitrev(ih, h)));
                has[h] = 1;
                ext_t rrow = {0, 0, 0, 0}, cur = {1, 0, 0, 0}, t;
                ext_t *apow = (ext_t *)malloc(W * sizeof(ext_t));
                for (size_t k = 0; k < W; k++) {
                    ext_set(apow[k], cur);
                    ext_scale(cur, rows[k], t);
                    ext_add(rrow, t, rrow);
                    ora_ext_mul(cur, alpha_f, cur);
                }
                for (unsigned pt = 0; pt < M->n_pts; pt++) {
                    ext_t z, ry = {0, 0, 0, 0}, off, d, di, num, u;
                    if (pt == 0) ext_set(z, zeta);
                    else ext_scale(zeta, ora_two_adic_generator(M->lh), z);
                    for (size_t k = 0; k < W; k++) {
                        ora_ext_mul(apow[k], opened[oi + k], t);
                        ext_add(ry, t, ry);
                    }
                    ext_pow(alpha_f, num_reduced[h], off);
                    ext_set(d, z);
                    d[0] = ora_sub(d[0], x);
                    ora_ext_inv(d, di);
                    ext_sub(ry, rrow, num);
                    ora_ext_mul(num, di, u);
                    ora_ext_mul(u, off, u);
                    ext_add(roq[h], u, roq[h]);
                    num_reduced[h] += W;
                    oi += W;
                }
                free(apow);
                rows += W;
            }
        }
        /* fold chain */
        ext_t eval;
        ext_set(eval, roq[hmax]);
        for (unsigned l = 0; l < n_layers; l++) {
            const unsigned log_len = hmax - l;
            size_t il = idx >> l, n_path = 8 * (size_t)(log_len - 1);
            if (r + 4 + n_path > n_words) { rc = -3; break; }
            const uint32_t *sib = proof + r, *path = proof + r + 4;
            r += 4 + n_path;
            uint32_t opening[8 + 8 * 32];
            memcpy(opening + 4 * (il & 1), eval, 16);
            memcpy(opening + 4 * ((il & 1) ^ 1), sib, 16);
            memcpy(opening + 8, path, n_path * 4);
            unsigned lhl = log_len - 1;
            size_t wl = 8;
            if (!ora_mmcs_verify(fri_hdr + 9 * l, &lhl, &wl, 1, il >> 1, opening)) { rc = -7; break; }
            ext_t folded;
            {   /* fold_row at index il>>1 of a layer with log height log_len-1 */
                uint32_t xx = ora_pow(ora_two_adic_generator(log_len), bitrev(il >> 1, log_len - 1));
                uint32_t c = ora_sub(0, ora_mul(ora_inv(xx), ora_inv(2)));
                ext_t d, bx, t;
                for (int k = 0; k < 4; k++) d[k] = ora_mul(ora_sub(opening[4 + k], opening[k]), c);
                ext_set(bx, betas[l]);
                bx[0] = ora_sub(bx[0], xx);
                ora_ext_mul(bx, d, t);
                for (int k = 0; k < 4; k++) folded[k] = ora_add(opening[k], t[k]);
            }
            ext_set(eval, folded);
            if (has[log_len - 1]) {
                ext_t b2, t;
                ora_ext_mul(betas[l], betas[l], b2);
                ora_ext_mul(b2, roq[log_len - 1], t);
                ext_add(eval, t, eval);
            }
        }
        if (rc == 0) { /* the folded value must be the final polynomial at this query's point of the last domain (Horner) */
            ext_t want = {0, 0, 0, 0};
            const uint32_t xf = lfp ? ora_pow(ora_two_adic_generator(b + lfp), bitrev(idx >> n_layers, b + lfp)) : 0;
            for (size_t j = n_fin; j-- > 0;) {
                for (int k = 0; k < 4; k++) want[k] = ora_add(ora_mul(want[k], xf), fin[4 * j + k]);
            }
            if (memcmp(eval, want, 16)) rc = -8;
        }
    }
    if (rc == 0 && r != n_words) rc = -3;
    free(betas), free(lhs_all), free(ws_all), free(roq), free(has), free(num_reduced), free(open_off), free(cm), free(qoff);
    free(batch_first), free(batch_n), free(batch_root), free(rows_of), free(prep_roots), free(has_prep);
    for (size_t a = 0; a < n_airs; a++) free_program(&pg[a]);
    free(pg);
    return rc;
}
