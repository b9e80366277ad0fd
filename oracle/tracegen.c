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
