// transcript.hpp -- device-resident duplex challenger state and its stream-ordered operations.
#pragma once
#include <stdint.h>

struct zkhip_ctx;

namespace zk {

// Lives in HBM; mutated only by single-lane kernels on the prover's stream.
// Pending inputs are written straight into the front of `state` (nothing reads the state between
// an observe and the next duplexing), and the output buffer is state[0..8] itself.
struct DevTranscript {
    uint32_t state[16];
    uint32_t n_in;
    uint32_t n_out;
    uint32_t pow_found;    // smallest satisfying witness of the current grind, or 0xffffffff
    uint32_t pow_applied;  // set once the witness has been absorbed
    uint32_t error;        // bit0: PoW search exhausted, bit1: internal inconsistency
    uint32_t pad;
};

int transcript_init(zkhip_ctx* ctx, DevTranscript* d_t);
// d_src: device words; canonical=true converts to Montgomery while absorbing
int transcript_observe(zkhip_ctx* ctx, DevTranscript* d_t, const uint32_t* d_src, uint32_t n, bool canonical);
// writes n sampled elements to d_monty (Montgomery) and/or d_canon (canonical); either may be null
int transcript_sample(zkhip_ctx* ctx, DevTranscript* d_t, uint32_t* d_monty, uint32_t* d_canon, uint32_t n);
int transcript_sample_bits(zkhip_ctx* ctx, DevTranscript* d_t, uint32_t* d_dst, uint32_t n, unsigned bits);
// d_witness_out: optional device word receiving the canonical witness
int transcript_grind(zkhip_ctx* ctx, DevTranscript* d_t, unsigned bits, uint32_t* d_witness_out);
// one FRI commit round: observe the layer's root (8 Montgomery words on the device), grind, sample the folding challenge -- ONE launch;
// d_proof_out receives [root (8, canonical) | witness], d_beta_out the challenge (4 words, Montgomery)
int transcript_fri_round(zkhip_ctx* ctx, DevTranscript* d_t, const uint32_t* d_root, unsigned bits, uint32_t* d_proof_out, uint32_t* d_beta_out);

}  // namespace zk
