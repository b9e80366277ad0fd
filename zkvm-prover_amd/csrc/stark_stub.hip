// temporary: STARK entry points not yet implemented (replaced by prover.hip / verifier.hip)
#include "zkhip_internal.hpp"
extern "C" {
int zkhip_keygen(zkhip_ctx* ctx, const zkhip_params*, const zkhip_air*, size_t, zkhip_pk**) { return zk::set_error(ctx, ZKHIP_ERR_INVALID, "not implemented"); }
void zkhip_pk_destroy(zkhip_ctx*, zkhip_pk*) {}
size_t zkhip_proof_size(const zkhip_pk*) { return 0; }
int zkhip_prove(zkhip_ctx* ctx, const zkhip_pk*, const uint32_t* const*, const uint32_t* const*, uint8_t*, size_t, size_t*) { return zk::set_error(ctx, ZKHIP_ERR_INVALID, "not implemented"); }
int zkhip_prove_async(zkhip_ctx* ctx, const zkhip_pk*, const uint32_t* const*, const uint32_t* const*) { return zk::set_error(ctx, ZKHIP_ERR_INVALID, "not implemented"); }
int zkhip_proof_fetch(zkhip_ctx* ctx, const zkhip_pk*, uint8_t*, size_t, size_t*) { return zk::set_error(ctx, ZKHIP_ERR_INVALID, "not implemented"); }
int zkhip_verify(const zkhip_params*, const zkhip_air*, size_t, const uint32_t* const*, const uint8_t*, size_t) { return ZKHIP_ERR_INVALID; }
}
