// svi_vae.hip — Wan 3-D causal VAE on gfx950 (placeholder until the conv kernels land: every entry
// point fails loudly with SVI_ERR_UNSUPPORTED; there is no CPU fallback).
#include "svi_common.h"

struct svi_vae { int dummy; };

extern "C" svi_status svi_vae_create(svi_vae** out) {
    SVI_REQUIRE(out, "svi_vae_create: null argument");
    *out = new (std::nothrow) svi_vae();
    if (!*out) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    return SVI_OK;
}
extern "C" svi_status svi_vae_destroy(svi_vae* h) { delete h; return SVI_OK; }
extern "C" svi_status svi_vae_bind_weight(svi_vae*, const char*, const void*, svi_dtype, const int64_t*, int32_t) {
    svi_set_error("VAE kernels not built yet"); return SVI_ERR_UNSUPPORTED;
}
extern "C" svi_status svi_vae_check_bound(svi_vae*) { svi_set_error("VAE kernels not built yet"); return SVI_ERR_UNSUPPORTED; }
extern "C" svi_status svi_vae_decode(svi_vae*, const float*, float*, int32_t, int32_t, int32_t, svi_stream) {
    svi_set_error("VAE kernels not built yet"); return SVI_ERR_UNSUPPORTED;
}
extern "C" svi_status svi_vae_encode(svi_vae*, const float*, float*, int32_t, int32_t, int32_t, svi_stream) {
    svi_set_error("VAE kernels not built yet"); return SVI_ERR_UNSUPPORTED;
}
