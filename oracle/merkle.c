/*
 * oracle/merkle.c -- Poseidon2 Merkle commitment over column-major matrices
 * of mixed power-of-two heights.  TEST INFRASTRUCTURE.  Pinned to the reference's stored proofs (72 batch openings
 * incl. 17-matrix mixed-height batches, tests/test_ref_vectors_cpu.py).
 *
 * Restates the published p3 MerkleTreeMmcs algorithm (the MMCS the
 * north_star's "Merkle-Poseidon2 commit" refers to; the crate itself is not
 * even in the reference's Cargo.lock -- SURVEY.md finding 2 -- so this follows
 * the v1-era construction SURVEY.md A.3 names):
 *   leaf digest i  = sponge(concat of row i of every tallest matrix);
 *   parent         = compress(left, right);
 *   when a layer of size s is produced and matrices of height s exist,
 *   node i         = compress(node i, sponge(concat of their rows i)).
 * Matrices are taken tallest first, ties in caller order.  An opening = the
 * opened rows (caller order) followed by log_height sibling digests
 * bottom-up.  Digest = 8 words (crates/types/src/proof.rs:209).
 */
#include <stdlib.h>
#include <string.h>
#include "zk_oracle.h"

struct ora_tree {
    unsigned log_height;
    size_t n_mats;
    ora_matrix *mats;
    uint32_t **layers; /* layers[l] = 8 * 2^(log_height-l) words */
};

static void hash_rows(const ora_matrix *mats, size_t n_mats, unsigned log_h, size_t row,
                      uint32_t *tmp, uint32_t out[8]) {
    size_t len = 0;
    for (size_t m = 0; m < n_mats; m++) {
        if (mats[m].log_height != log_h) continue;
        for (size_t c = 0; c < mats[m].width; c++) tmp[len++] = mats[m].data[c * mats[m].stride + row];
    }
    ora_hash_slice(tmp, len, out);
}

ora_tree *ora_mmcs_commit(const ora_matrix *mats, size_t n_mats, uint32_t root[8]) {
    ora_tree *t = (ora_tree *)calloc(1, sizeof *t);
    unsigned lh = 0;
    size_t total_w = 0;
    for (size_t m = 0; m < n_mats; m++) {
        if (mats[m].log_height > lh) lh = mats[m].log_height;
        total_w += mats[m].width;
    }
    t->log_height = lh;
    t->n_mats = n_mats;
    t->mats = (ora_matrix *)malloc(n_mats * sizeof(ora_matrix));
    memcpy(t->mats, mats, n_mats * sizeof(ora_matrix));
    t->layers = (uint32_t **)calloc(lh + 1, sizeof(uint32_t *));
    size_t n = (size_t)1 << lh;
    t->layers[0] = (uint32_t *)malloc(n * 8 * sizeof(uint32_t));
#pragma omp parallel
    {
        uint32_t *tmp = (uint32_t *)malloc((total_w + 1) * sizeof(uint32_t));
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; i++) hash_rows(mats, n_mats, lh, i, tmp, t->layers[0] + 8 * i);
        free(tmp);
    }
    for (unsigned l = 1; l <= lh; l++) {
        size_t cnt = (size_t)1 << (lh - l);
        unsigned cur_log = lh - l;
        int inject = 0;
        for (size_t m = 0; m < n_mats; m++)
            if (mats[m].log_height == cur_log) inject = 1;
        t->layers[l] = (uint32_t *)malloc(cnt * 8 * sizeof(uint32_t));
        const uint32_t *prev = t->layers[l - 1];
#pragma omp parallel
        {
            uint32_t *tmp = (uint32_t *)malloc((total_w + 1) * sizeof(uint32_t));
#pragma omp for schedule(static)
            for (size_t i = 0; i < cnt; i++) {
                uint32_t *o = t->layers[l] + 8 * i;
                ora_compress(prev + 16 * i, prev + 16 * i + 8, o);
                if (inject) {
                    uint32_t h[8];
                    hash_rows(mats, n_mats, cur_log, i, tmp, h);
                    ora_compress(o, h, o);
                }
            }
            free(tmp);
        }
    }
    memcpy(root, t->layers[lh], 8 * sizeof(uint32_t));
    return t;
}

unsigned ora_tree_log_height(const ora_tree *t) { return t->log_height; }
const uint32_t *ora_tree_layer(const ora_tree *t, unsigned layer) { return t->layers[layer]; }

size_t ora_mmcs_open(const ora_tree *t, size_t index, uint32_t *out) {
    size_t w = 0;
    for (size_t m = 0; m < t->n_mats; m++) {
        const ora_matrix *M = &t->mats[m];
        size_t row = index >> (t->log_height - M->log_height);
        for (size_t c = 0; c < M->width; c++) out[w++] = M->data[c * M->stride + row];
    }
    for (unsigned l = 0; l < t->log_height; l++) {
        size_t sib = (index >> l) ^ 1;
        memcpy(out + w, t->layers[l] + 8 * sib, 32);
        w += 8;
    }
    return w;
}

int ora_mmcs_verify(const uint32_t root[8], const unsigned *log_heights, const size_t *widths,
                    size_t n_mats, size_t index, const uint32_t *opening) {
    unsigned lh = 0;
    size_t total_w = 0;
    for (size_t m = 0; m < n_mats; m++) {
        if (log_heights[m] > lh) lh = log_heights[m];
        total_w += widths[m];
    }
    uint32_t *tmp = (uint32_t *)malloc((total_w + 1) * sizeof(uint32_t));
    const uint32_t *path = opening + total_w;
    uint32_t cur[8];
    /* rows of matrices with a given log height, concatenated in caller order */
    for (unsigned level = lh;; level--) {
        size_t len = 0, off = 0;
        int any = 0;
        for (size_t m = 0; m < n_mats; m++) {
            if (log_heights[m] == level) {
                memcpy(tmp + len, opening + off, widths[m] * 4);
                len += widths[m];
                any = 1;
            }
            off += widths[m];
        }
        if (level == lh) {
            ora_hash_slice(tmp, len, cur);
        } else {
            unsigned l = lh - level - 1; /* sibling layer index consumed to get here */
            const uint32_t *sib = path + 8 * l;
            if (((index >> l) & 1) == 0) ora_compress(cur, sib, cur);
            else ora_compress(sib, cur, cur);
            if (any) {
                uint32_t h[8];
                ora_hash_slice(tmp, len, h);
                ora_compress(cur, h, cur);
            }
        }
        if (level == 0) break;
    }
    free(tmp);
    return memcmp(cur, root, 32) == 0;
}

void ora_tree_free(ora_tree *t) {
    if (!t) return;
    for (unsigned l = 0; l <= t->log_height; l++) free(t->layers[l]);
    free(t->layers);
    free(t->mats);
    free(t);
}

/* Test utility for tests/golden/gen_ref_vectors.py: the query index is not part of a reference proof (it comes from
 * the transcript), so it is recovered by trying every leaf position of a single-matrix opening.  Walks the path
 * bottom-up keeping the node of every candidate prefix (2^(l) candidates after l siblings): 2^(log_height+1)
 * compressions in all.  Returns how many indices verify; the first goes to *index_out. */
size_t ora_mmcs_find_index(const uint32_t root[8], unsigned log_height, size_t width, const uint32_t *opening,
                           size_t *index_out) {
    size_t n = (size_t)1 << log_height;
    uint32_t *cur = (uint32_t *)malloc(n * 8 * sizeof(uint32_t)), *nxt = (uint32_t *)malloc(n * 8 * sizeof(uint32_t));
    const uint32_t *path = opening + width;
    ora_hash_slice(opening, width, cur);
    for (unsigned l = 0; l < log_height; l++) {
        size_t cnt = (size_t)1 << l;
        const uint32_t *sib = path + 8 * l;
#pragma omp parallel for schedule(static)
        for (size_t c = 0; c < cnt; c++) {
            ora_compress(cur + 8 * c, sib, nxt + 8 * c);         /* bit l of the index = 0 */
            ora_compress(sib, cur + 8 * c, nxt + 8 * (c + cnt)); /* bit l of the index = 1 */
        }
        uint32_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    size_t found = 0;
    for (size_t c = 0; c < n; c++)
        if (memcmp(cur + 8 * c, root, 32) == 0 && found++ == 0) *index_out = c;
    free(cur);
    free(nxt);
    return found;
}
