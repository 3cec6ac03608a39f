// svi_dit.hip — the Wan DiT denoiser forward as ONE C call per forward: weight table bound by
// reference state-dict key, static workspace in HBM, every kernel enqueued on the caller's stream.
//
// Stands in for model_fn_wan_video (pipelines/svi_video.py:74-137) == WanModel.forward
// (models/wan_video_dit.py:486-567) and DiTBlock.forward (:354-374).
//
// HBM layout of one forward (bf16 unless noted; L = f*h*w tokens, D = dim, F = ffn_dim):
//   X   [L, D]      residual stream (updated in place by the GEMM epilogues)
//   Hb  [L, D]      normed+modulated GEMM input, then attention output
//   QK  [L, 2D]     q | k, token-major; RMSNorm+RoPE in place
//   VT  [D, L8]     V transposed (channel-major), emitted directly by the swapped V projection GEMM
//   Fb  [L, F]      FFN hidden (GELU applied in the producing GEMM's epilogue)
//   CTX [Lc(+257), D], CK [.., D], CVT [D, ..]   projected context and its per-block K / V^T
//   modf f32 [layers][6][D]   (modulation + t_mod), bf16-rounded, with (1+scale) pre-added
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include <math.h>
#include <string.h>

#include "svi_common.h"

namespace {

struct Lin { const bf16* w = nullptr; const bf16* b = nullptr; };
struct AttnW {
    Lin q, k, v, o;
    const bf16* norm_q = nullptr;
    const bf16* norm_k = nullptr;
    Lin k_img, v_img;
    const bf16* norm_k_img = nullptr;
};
struct BlockW {
    const bf16* modulation = nullptr;
    AttnW sa, ca;
    const bf16* norm3_w = nullptr;
    const bf16* norm3_b = nullptr;
    Lin ffn0, ffn2;
    const unsigned char* proj_8[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // sa.q, sa.k, sa.v, sa.o, ca.q, ca.o as stored e4m3 bytes (svi_dit_proj_mx8)
    const unsigned char *ffn0_8 = nullptr, *ffn2_8 = nullptr;      // the same weights as stored e4m3 bytes (FP8 storage mode), for the opt-in MX-fp8 MLP
    // talk variant (enable_multitalk): audio cross-attention (models/attention.py:283-371) and its pre-norm (dit:351)
    Lin aud_q, aud_kv, aud_proj;
    const bf16* normx_w = nullptr;
    const bf16* normx_b = nullptr;
};
struct Slot { const bf16** ptr; std::vector<int64_t> shape; };

struct Workspace {
    int L = 0, Lc = 0;              // the problem the pointers below are laid out for
    int capL = 0, capLc = 0;        // what the allocation holds (grow only)
    char* base = nullptr;
    size_t bytes = 0;
    bf16 *X, *X2, *Hb, *QK, *VT, *Fb, *CTX, *CTXH, *CK, *CVT, *CKi, *CVTi, *A2, *PATCH, *HO, *IMG0, *IMG1, *IMGD;
    bf16 *e, *h1, *t, *st, *tmod;
    float *modf, *headf;
    float *RSS = nullptr, *RS = nullptr;      // cross-attention q: per-64-column row sums of squares [dim / 64][ldss] from the projection's epilogue, rs [L]
    int ldss = 0;
    int* tail = nullptr;            // identical-suffix summary of the text context when no cache entry holds it
    unsigned char* Q8 = nullptr;    // MX-fp8 MLP: e4m3 activations [L, max(D, F)] and their block scales [F/128][sc_rows]
    unsigned* S8 = nullptr;
    int sc_rows = 0;
    int ldvt = 0, ldcvt = 0, ldcvti = 264, kpatch = 0;
};

}  // namespace

// A projected context (text_embedding / img_emb output) and the cross-attention K / V^T of every block derived from it.
// These depend only on (prompt embedding, CLIP feature, weights): the reference recomputes them in every one of the 100
// forwards of a clip ("hoistable"); with svi_dit_context_cache(h, 1) they are computed once per distinct
// context pointer and reused until svi_dit_context_cache(h, 0) / a re-bind invalidates them.
struct CtxEntry {
    const void* key_ctx = nullptr;
    const void* key_clip = nullptr;
    int Lc = 0;
    char* base = nullptr;
    bf16* CTX = nullptr;
    std::vector<bf16*> CK, CVT, CKi, CVTi;
    int* tail = nullptr;            // {effective key count, multiplicity of the last effective key}: see ctx_tail_*_kernel
    int key_blocks = 0;             // host copy of ceil(tail[0] / 32): how many 32-key blocks the cross-attention walks; 0 = not read (filled under capture)
    bool filled = false;
    unsigned long long stamp = 0;
};

struct CtxKV { bf16 *CK, *CVT, *CKi, *CVTi; bool compute; const int* tail; int key_blocks; };

struct svi_dit {
    svi_dit_config cfg;
    int device = -1;                  // claimed by the first compute call (svi_claim_device)
    std::vector<BlockW> blocks;
    const bf16 *patch_w = nullptr, *patch_b = nullptr;
    Lin text0, text2, time0, time2, timeproj, head;
    const bf16* head_mod = nullptr;
    const bf16 *img_ln0_w = nullptr, *img_ln0_b = nullptr, *img_ln4_w = nullptr, *img_ln4_b = nullptr;
    Lin img1, img3;
    std::map<std::string, Slot> slots;
    Workspace ws;
    // rope
    int rf = 0, rh = 0, rw = 0;
    float2* rope_dev = nullptr;
    SviRope rope{};
    // bumped whenever a device pointer a captured hipGraph may have baked in stops being valid (workspace / context-entry
    // (re)allocation, context-cache reset, re-bind): svi_dit_generation
    unsigned long long generation = 0;
    // forward_pair: the smallest stacked row count (2 L) whose doubled workspace did not fit.  Sticky for the handle's life (the workspace only grows, and
    // the memory that was missing belongs to the caller's weights): later steps of that size go straight to the unstacked form instead of freeing and
    // re-allocating per step — and a capture pass behind the eager step no longer meets "the workspace must grow while captured" (ADVICE r5).
    int pair_stack_oom_rows = 0;
    bool ffn_mx8 = false;             // opt-in: the MLP GEMMs on the block-scaled fp8 matrix path (svi_dit_ffn_mx8)
    bool proj_mx8 = false;            // opt-in: the block's other six projections too (svi_dit_proj_mx8)
    // context cache
    bool ctx_cache_on = false;
    CtxEntry ctx_entries[4];
    unsigned long long ctx_clock = 0;
    // talk variant: audio projection weights (AudioProjModel, dit:44-115), the armed audio windows and the per-size audio workspace
    Lin ap1, ap1vf, ap2, ap3;
    const bf16 *ap_norm_w = nullptr, *ap_norm_b = nullptr;
    const bf16 *aud_first = nullptr, *aud_latter = nullptr;
    int aud_latter_n = 0;
    char* aud_base = nullptr;
    int aud_frames = 0;
    bf16 *AH0 = nullptr, *AH1 = nullptr, *AP3 = nullptr, *AUD = nullptr, *AK = nullptr, *AVT = nullptr;
    int ldavt = 0;
    // sequence-parallel shard in flight (svi_dit_sp_begin .. svi_dit_sp_head)
    int sp_rows = 0, sp_row0 = 0, sp_Lc = 0;
    int sp_nb = 1;                    // 2: the shard carries both CFG branches stacked (svi_dit_sp_begin_pair): X = [cond rows | uncond rows]
    bool sp_active = false;
    const bf16* sp_ctxp = nullptr;
    const bf16* sp_ctxp_b = nullptr;
    std::vector<CtxKV> sp_kv, sp_kv_b;
};

// ------------------------------------------------------------------------------------------------
// small kernels private to the model forward
// ------------------------------------------------------------------------------------------------
// sinusoidal_embedding_1d (models/wan_video_dit.py:154-158): fp64 angle, cos half | sin half, -> bf16
__global__ void sinusoid_kernel(const float* __restrict__ timestep, bf16* __restrict__ e, int freq_dim) {
    const int half = freq_dim >> 1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    const double pos = (double)timestep[0];
    const double ang = pos * pow(10000.0, -((double)j / (double)half));
    e[j] = (bf16)(float)cos(ang);
    e[half + j] = (bf16)(float)sin(ang);
}

__global__ void silu_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bf16)silu_f((float)in[i]);
}

// patch tokens: P[l][k], k = ((c*pt + a)*ph + b)*pw + cw ; token l = (fi*hh + hi)*ww + wi  (dit:473-477)
__global__ void patch_gather_kernel(const bf16* __restrict__ x, const bf16* __restrict__ y, bf16* __restrict__ P,
                                    int Cx, int Cy, int T, int H, int W, int pt, int ph, int pw, int kp, int L, int row0) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)L * kp) return;
    const int lr = (int)(idx / kp), k = (int)(idx - (int64_t)lr * kp);
    const int l = lr + row0;                         // token index in the (f h w) grid; row lr of this launch
    const int hh = H / ph, ww = W / pw;
    const int K = (Cx + Cy) * pt * ph * pw;
    if (k >= K) { P[idx] = (bf16)0.f; return; }
    const int cw = k % pw, b = (k / pw) % ph, a = (k / (pw * ph)) % pt, c = k / (pw * ph * pt);
    const int wi = l % ww, hi = (l / ww) % hh, fi = l / (ww * hh);
    const size_t sp = ((size_t)(fi * pt + a) * H + (hi * ph + b)) * W + (wi * pw + cw);
    P[idx] = (c < Cx) ? x[(size_t)c * T * H * W + sp] : y[(size_t)(c - Cx) * T * H * W + sp];
}

// unpatchify 'b (f h w) (x y z c) -> b c (f x) (h y) (w z)'  (dit:479-484)
__global__ void unpatchify_kernel(const bf16* __restrict__ ho, bf16* __restrict__ out, int C, int T, int H, int W,
                                  int pt, int ph, int pw, int ldho) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)C * T * H * W;
    if (idx >= n) return;
    const int wq = (int)(idx % W), hq = (int)((idx / W) % H), tq = (int)((idx / ((int64_t)W * H)) % T);
    const int c = (int)(idx / ((int64_t)W * H * T));
    const int hh = H / ph, ww = W / pw;
    const int fi = tq / pt, a = tq % pt, hi = hq / ph, b = hq % ph, wi = wq / pw, cw = wq % pw;
    const int l = (fi * hh + hi) * ww + wi;
    const int col = ((a * ph + b) * pw + cw) * C + c;
    out[idx] = ho[(size_t)l * ldho + col];
}

// ------------------------------------------------------------------------------------------------
// handle construction / weight binding
// ------------------------------------------------------------------------------------------------
static void add_slot(svi_dit* h, const std::string& name, const bf16** p, std::vector<int64_t> shape) {
    h->slots[name] = Slot{p, std::move(shape)};
}
static void add_lin(svi_dit* h, const std::string& name, Lin* l, int64_t o, int64_t i) {
    add_slot(h, name + ".weight", &l->w, {o, i});
    add_slot(h, name + ".bias", &l->b, {o});
}

extern "C" svi_status svi_dit_create(const svi_dit_config* c, svi_dit** out) {
    SVI_REQUIRE(c && out, "svi_dit_create: null argument");
    SVI_REQUIRE(c->num_heads > 0 && c->dim == c->num_heads * 128, "head_dim must be 128 (dim=%d heads=%d)", c->dim,
                c->num_heads);
    SVI_REQUIRE(c->dim % 8 == 0 && c->ffn_dim % 8 == 0 && c->text_dim % 8 == 0 && c->freq_dim % 8 == 0,
                "dim/ffn_dim/text_dim/freq_dim must be multiples of 8");
    SVI_REQUIRE(c->patch_t > 0 && c->patch_h > 0 && c->patch_w > 0 && c->num_layers > 0, "bad patch/layers");
    SVI_REQUIRE((c->in_dim * c->patch_t * c->patch_h * c->patch_w) % 8 == 0, "in_dim*prod(patch) must be a multiple of 8");
    SVI_REQUIRE(c->has_image_input ? c->in_dim > 16 : true, "has_image_input needs in_dim > 16");
    svi_dit* h = new (std::nothrow) svi_dit();
    if (!h) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    h->cfg = *c;
    const int64_t D = c->dim, F = c->ffn_dim;
    h->blocks.resize(c->num_layers);
    add_slot(h, "patch_embedding.weight", &h->patch_w, {D, c->in_dim, c->patch_t, c->patch_h, c->patch_w});
    add_slot(h, "patch_embedding.bias", &h->patch_b, {D});
    add_lin(h, "text_embedding.0", &h->text0, D, c->text_dim);
    add_lin(h, "text_embedding.2", &h->text2, D, D);
    add_lin(h, "time_embedding.0", &h->time0, D, c->freq_dim);
    add_lin(h, "time_embedding.2", &h->time2, D, D);
    add_lin(h, "time_projection.1", &h->timeproj, 6 * D, D);
    for (int i = 0; i < c->num_layers; ++i) {
        BlockW& b = h->blocks[i];
        const std::string p = "blocks." + std::to_string(i) + ".";
        add_slot(h, p + "modulation", &b.modulation, {1, 6, D});
        for (int a = 0; a < 2; ++a) {
            AttnW& w = a ? b.ca : b.sa;
            const std::string ap = p + (a ? "cross_attn." : "self_attn.");
            add_lin(h, ap + "q", &w.q, D, D);
            add_lin(h, ap + "k", &w.k, D, D);
            add_lin(h, ap + "v", &w.v, D, D);
            add_lin(h, ap + "o", &w.o, D, D);
            add_slot(h, ap + "norm_q.weight", &w.norm_q, {D});
            add_slot(h, ap + "norm_k.weight", &w.norm_k, {D});
            if (a && c->has_image_input) {
                add_lin(h, ap + "k_img", &w.k_img, D, D);
                add_lin(h, ap + "v_img", &w.v_img, D, D);
                add_slot(h, ap + "norm_k_img.weight", &w.norm_k_img, {D});
            }
        }
        add_slot(h, p + "norm3.weight", &b.norm3_w, {D});
        add_slot(h, p + "norm3.bias", &b.norm3_b, {D});
        add_lin(h, p + "ffn.0", &b.ffn0, F, D);
        add_lin(h, p + "ffn.2", &b.ffn2, D, F);
        if (c->enable_multitalk) {          // dit:338-351: encoder_hidden_states_dim = 768, qkv_bias, no qk norm
            add_lin(h, p + "audio_cross_attn.q_linear", &b.aud_q, D, D);
            add_lin(h, p + "audio_cross_attn.kv_linear", &b.aud_kv, 2 * D, 768);
            add_lin(h, p + "audio_cross_attn.proj", &b.aud_proj, D, D);
            add_slot(h, p + "norm_x.weight", &b.normx_w, {D});
            add_slot(h, p + "norm_x.bias", &b.normx_b, {D});
        }
    }
    if (c->enable_multitalk) {              // dit:455-470: audio_window 5, vae_scale 4, 12 blocks x 768 channels, 512 hidden, 32 tokens x 768
        add_lin(h, "audio_proj.proj1", &h->ap1, 512, 5 * 12 * 768);
        add_lin(h, "audio_proj.proj1_vf", &h->ap1vf, 512, 8 * 12 * 768);
        add_lin(h, "audio_proj.proj2", &h->ap2, 512, 512);
        add_lin(h, "audio_proj.proj3", &h->ap3, 32 * 768, 512);
        add_slot(h, "audio_proj.norm.weight", &h->ap_norm_w, {768});
        add_slot(h, "audio_proj.norm.bias", &h->ap_norm_b, {768});
    }
    add_slot(h, "head.modulation", &h->head_mod, {1, 2, D});
    add_lin(h, "head.head", &h->head, (int64_t)c->out_dim * c->patch_t * c->patch_h * c->patch_w, D);
    if (c->has_image_input) {
        add_slot(h, "img_emb.proj.0.weight", &h->img_ln0_w, {1280});
        add_slot(h, "img_emb.proj.0.bias", &h->img_ln0_b, {1280});
        add_lin(h, "img_emb.proj.1", &h->img1, 1280, 1280);
        add_lin(h, "img_emb.proj.3", &h->img3, D, 1280);
        add_slot(h, "img_emb.proj.4.weight", &h->img_ln4_w, {D});
        add_slot(h, "img_emb.proj.4.bias", &h->img_ln4_b, {D});
    }
    *out = h;
    return SVI_OK;
}

extern "C" svi_status svi_dit_destroy(svi_dit* h) {
    if (!h) return SVI_OK;
    if (h->ws.base) (void)hipFree(h->ws.base);
    if (h->aud_base) (void)hipFree(h->aud_base);
    if (h->rope_dev) (void)hipFree(h->rope_dev);
    for (auto& e : h->ctx_entries)
        if (e.base) (void)hipFree(e.base);
    delete h;
    return SVI_OK;
}

extern "C" svi_status svi_dit_bind_weight(svi_dit* h, const char* name, const void* dev_ptr, svi_dtype dtype,
                                          const int64_t* shape, int32_t rank) {
    SVI_REQUIRE(h && name && dev_ptr && shape, "svi_dit_bind_weight: null argument");
    auto it = h->slots.find(name);
    if (it == h->slots.end()) { svi_set_error("unknown DiT parameter '%s'", name); return SVI_ERR_INVALID; }
    SVI_REQUIRE(dtype == SVI_BF16, "DiT parameter '%s' must be bf16", name);
    const std::vector<int64_t>& want = it->second.shape;
    bool ok = (int)want.size() == rank;
    for (int i = 0; ok && i < rank; ++i) ok = want[i] == shape[i];
    if (!ok) { svi_set_error("shape mismatch for '%s'", name); return SVI_ERR_INVALID; }
    SVI_REQUIRE(((uintptr_t)dev_ptr % 16) == 0, "parameter '%s' is not 16-byte aligned", name);
    *it->second.ptr = reinterpret_cast<const bf16*>(dev_ptr);
    for (auto& e : h->ctx_entries) e.filled = false;      // cached projections were made with the old weights
    ++h->generation;
    return SVI_OK;
}

// Opt-in MX-fp8 MLP (north_star "bf16/fp8 MFMA"; the reference has no fp8 arithmetic, only e4m3 weight STORAGE: test_svi.py:337,
// vram_management/layers.py:65-71).  svi_dit_bind_ffn_fp8 hands over the stored e4m3 bytes of blocks.<layer>.ffn.<0|2>.weight — the
// very values whose bf16 casts are bound as the ordinary weights — and svi_dit_ffn_mx8(h, 1) routes both MLP GEMMs of every block
// through csrc/svi_gemm.hip's block-scaled fp8 kernel (activations quantised per row and 32-element block).  Arithmetic the reference
// never performs: separately toleranced (tests/test_gpu_mx8.py), a separate bench line, never the default.
extern "C" svi_status svi_dit_bind_ffn_fp8(svi_dit* h, int32_t layer, int32_t which, const void* e4m3_weight) {
    SVI_REQUIRE(h && e4m3_weight && layer >= 0 && layer < h->cfg.num_layers && (which == 0 || which == 2 || (which >= 10 && which <= 15)), "svi_dit_bind_ffn_fp8: bad argument");
    SVI_REQUIRE(((uintptr_t)e4m3_weight % 16) == 0, "svi_dit_bind_ffn_fp8: the weight is not 16-byte aligned");
    if (which >= 10) h->blocks[layer].proj_8[which - 10] = reinterpret_cast<const unsigned char*>(e4m3_weight);      // 10..15: sa.q, sa.k, sa.v, sa.o, ca.q, ca.o
    else (which == 0 ? h->blocks[layer].ffn0_8 : h->blocks[layer].ffn2_8) = reinterpret_cast<const unsigned char*>(e4m3_weight);
    return SVI_OK;
}
// The block's other six projections on the same path (ABI v10; include/svi_hip.h).
extern "C" svi_status svi_dit_proj_mx8(svi_dit* h, int32_t enable) {
    SVI_REQUIRE(h, "null handle");
    if (enable) {
        SVI_REQUIRE(h->cfg.dim % 256 == 0, "svi_dit_proj_mx8: dim must be a multiple of 256 (dim = %d)", h->cfg.dim);
        for (int l = 0; l < h->cfg.num_layers; ++l)
            for (int i = 0; i < 6; ++i)
                if (!h->blocks[l].proj_8[i]) { svi_set_error("svi_dit_proj_mx8: blocks.%d projection %d was not bound as e4m3 (svi_dit_bind_ffn_fp8, which = %d)", l, i, 10 + i); return SVI_ERR_UNBOUND; }
    }
    h->proj_mx8 = enable != 0;
    ++h->generation;
    return SVI_OK;
}
extern "C" svi_status svi_dit_ffn_mx8(svi_dit* h, int32_t enable) {
    SVI_REQUIRE(h, "null handle");
    if (enable) {
        SVI_REQUIRE(h->cfg.dim % 128 == 0 && h->cfg.ffn_dim % 128 == 0, "svi_dit_ffn_mx8: dim and ffn_dim must be multiples of 128");
        for (int l = 0; l < h->cfg.num_layers; ++l)
            if (!h->blocks[l].ffn0_8 || !h->blocks[l].ffn2_8) { svi_set_error("svi_dit_ffn_mx8: blocks.%d.ffn weights were not bound as e4m3 (svi_dit_bind_ffn_fp8)", l); return SVI_ERR_UNBOUND; }
    }
    h->ffn_mx8 = enable != 0;
    ++h->generation;
    return SVI_OK;
}

extern "C" svi_status svi_dit_context_cache(svi_dit* h, int32_t enable) {
    SVI_REQUIRE(h, "null handle");
    h->ctx_cache_on = enable != 0;
    for (auto& e : h->ctx_entries) e.filled = false;
    ++h->generation;
    return SVI_OK;
}

extern "C" svi_status svi_dit_check_bound(svi_dit* h) {
    SVI_REQUIRE(h, "null handle");
    for (auto& kv : h->slots)
        if (*kv.second.ptr == nullptr) { svi_set_error("parameter '%s' was never bound", kv.first.c_str()); return SVI_ERR_UNBOUND; }
    return SVI_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace and rope tables (allocation only when the problem size changes — never in steady state)
// ------------------------------------------------------------------------------------------------
static size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

// The allocation only grows: a problem that fits (tokens and context length both within what is held) is laid out inside the
// buffer that is there — mixing the stacked CFG pair (2 L rows) with plain forwards (L rows) on one handle, or prompts of two
// lengths, re-lays the pointers and re-zeroes the V^T pads on the stream, and neither frees nor allocates (legal under stream
// capture; an allocation is not, and is refused there with a message).
static svi_status ensure_workspace(svi_dit* h, int L, int Lc, hipStream_t st) {
    Workspace& w = h->ws;
    if (w.base && w.L == L && w.Lc == Lc && (w.Q8 != nullptr) == (h->ffn_mx8 || h->proj_mx8)) return SVI_OK;
    const svi_dit_config& c = h->cfg;
    const size_t D = c.dim, F = c.ffn_dim;
    const int img = c.has_image_input ? 257 : 0;
    const int kpatch = c.in_dim * c.patch_t * c.patch_h * c.patch_w;
    const size_t ho = (size_t)c.out_dim * c.patch_t * c.patch_h * c.patch_w;
    const size_t ho_ld = (ho + 7) / 8 * 8;
    size_t oX, oX2, oH, oQK, oVT, oF, oCTX, oCTXH, oCK, oCVT, oCKi, oCVTi, oA2, oP, oHO, oI0, oI1, oID, oe, oh1, ot, ost, otm, omodf, oheadf, otail, oRSS, oRS, oQ8 = 0, oS8 = 0;
    const bool mx8 = h->ffn_mx8 || h->proj_mx8;
    auto layout = [&](int l, int lc) -> size_t {
        const size_t Lctx = (size_t)lc + img;
        const int ldvt = ((l + 7) / 8) * 8, ldcvt = ((lc + 7) / 8) * 8;
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
        oX = take((size_t)l * D * 2); oX2 = take((size_t)l * D * 2); oH = take((size_t)l * D * 2); oQK = take((size_t)l * 2 * D * 2);
        oVT = take(D * ldvt * 2); oF = take((size_t)l * F * 2);
        oCTX = take(Lctx * D * 2); oCTXH = take((size_t)lc * D * 2); oCK = take(Lctx * D * 2);
        oCVT = take(D * ldcvt * 2); oCKi = take((size_t)264 * D * 2); oCVTi = take(D * 264 * 2);
        oA2 = take(img ? (size_t)l * D * 2 : 256);
        oP = take((size_t)l * kpatch * 2); oHO = take((size_t)l * ho_ld * 2);
        oI0 = take((size_t)264 * 1280 * 2); oI1 = take((size_t)264 * 1280 * 2); oID = take((size_t)264 * D * 2);
        oe = take(c.freq_dim * 2); oh1 = take(D * 2); ot = take(D * 2); ost = take(D * 2); otm = take(6 * D * 2);
        omodf = take((size_t)c.num_layers * 6 * D * 4); oheadf = take(2 * D * 4);
        otail = take((size_t)(lc + 8) * 4);
        oRSS = take((D / 64) * (size_t)ldvt * 4); oRS = take((size_t)ldvt * 4);
        if (mx8) { oQ8 = take((size_t)l * F); oS8 = take((size_t)(F / 128) * (size_t)(((l + 255) / 256) * 256) * 4); }
        return off;
    };
    const size_t need = layout(L, Lc);
    if (!w.base || need > w.bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            svi_set_error("the DiT workspace must grow (%zu B) while the stream is being captured: run this problem size once before the capture", need);
            return SVI_ERR_INVALID;
        }
        (void)hipGetLastError();
        const int gl = L > w.capL ? L : w.capL, glc = Lc > w.capLc ? Lc : w.capLc;
        size_t bytes = layout(gl, glc);
        if (bytes < need) bytes = need;
        if (svi_switches().ws_limit_mb && bytes > (size_t)svi_switches().ws_limit_mb << 20) {          // the caller's budget: refused BEFORE the workspace that is there is given up
            svi_set_error("DiT workspace of %zu B is beyond SVI_WS_LIMIT_MB=%d", bytes, svi_switches().ws_limit_mb);
            return SVI_ERR_OOM;
        }
        if (w.base) { SVI_CHECK_HIP(hipFree(w.base)); w.base = nullptr; w.bytes = 0; }
        hipError_t e = hipMalloc((void**)&w.base, bytes);
        if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B workspace) failed: %s", bytes, hipGetErrorString(e)); return SVI_ERR_OOM; }
        w.bytes = bytes; w.capL = gl; w.capLc = glc;
        (void)layout(L, Lc);
    }
    ++h->generation;                    // the pointers move (also inside an unchanged allocation): a captured graph of the old layout is stale
    // V^T pad columns (keys past the last one, up to the next multiple of 8) must read as zeros: the V^T regions are cleared whenever
    // the layout moves (stream-ordered; the scratch rows around them are rewritten by every forward before they are read)
    const int ldvt = ((L + 7) / 8) * 8, ldcvt = ((Lc + 7) / 8) * 8;
    SVI_CHECK_HIP(hipMemsetAsync(w.base + oVT, 0, al(D * ldvt * 2), st));
    SVI_CHECK_HIP(hipMemsetAsync(w.base + oCVT, 0, al(D * ldcvt * 2), st));
    SVI_CHECK_HIP(hipMemsetAsync(w.base + oCVTi, 0, al(D * 264 * 2), st));
    w.L = L; w.Lc = Lc; w.ldvt = ldvt; w.ldcvt = ldcvt; w.kpatch = kpatch;
    auto P = [&](size_t o) { return reinterpret_cast<bf16*>(w.base + o); };
    w.X = P(oX); w.X2 = P(oX2); w.Hb = P(oH); w.QK = P(oQK); w.VT = P(oVT); w.Fb = P(oF); w.CTX = P(oCTX); w.CTXH = P(oCTXH);
    w.CK = P(oCK); w.CVT = P(oCVT); w.CKi = P(oCKi); w.CVTi = P(oCVTi); w.A2 = P(oA2); w.PATCH = P(oP); w.HO = P(oHO);
    w.IMG0 = P(oI0); w.IMG1 = P(oI1); w.IMGD = P(oID); w.e = P(oe); w.h1 = P(oh1); w.t = P(ot); w.st = P(ost); w.tmod = P(otm);
    w.modf = reinterpret_cast<float*>(w.base + omodf);
    w.headf = reinterpret_cast<float*>(w.base + oheadf);
    w.tail = reinterpret_cast<int*>(w.base + otail);
    w.RSS = reinterpret_cast<float*>(w.base + oRSS); w.RS = reinterpret_cast<float*>(w.base + oRS); w.ldss = ldvt;
    w.Q8 = mx8 ? reinterpret_cast<unsigned char*>(w.base + oQ8) : nullptr;
    w.S8 = mx8 ? reinterpret_cast<unsigned*>(w.base + oS8) : nullptr;
    w.sc_rows = ((L + 255) / 256) * 256;
    return SVI_OK;
}

// ---- identical trailing rows of the text context -----------------------------------------------------------------------------
// The prompter zero-fills the rows of a prompt embedding past the prompt's own tokens (prompters/wan_prompter.py:107-108: 512 positions,
// a few dozen of them real), and cross-attention attends to all 512 without a mask (dit:266-303).  Every kernel between the embedding
// and the cross-attention K / V is row-local (text_embedding MLP, K / V projections, RMSNorm; no RoPE on the context), so identical
// input rows give identical K rows and identical V rows, and m identical keys are ONE key whose probability counts m times:
//   softmax over {s_0 .. s_{n-1}, s_n x m}  ==  softmax with the last score raised by ln m.
// ctx_tail finds the identical suffix of the INPUT rows (bitwise comparison with the last row — nothing is assumed about zeros) and
// leaves {n + 1, m} on the device; the cross-attention launch reads it there (no host round trip) and walks n + 1 keys instead of Lc.
// Per step at BASELINE configs[1]: 60 launches over 512 keys -> 60 launches over 65 / 33 keys.  SVI_CROSS_DEDUP=0 attends to every row.
__global__ void ctx_tail_rows_kernel(const bf16* __restrict__ ctx, int Lc, int dim, int* __restrict__ eq) {
    const int r = blockIdx.x;
    const u32x4* a = reinterpret_cast<const u32x4*>(ctx + (size_t)r * dim);
    const u32x4* b = reinterpret_cast<const u32x4*>(ctx + (size_t)(Lc - 1) * dim);
    int same = 1;
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) {
        const u32x4 x = a[i], y = b[i];
        same &= (x[0] == y[0]) & (x[1] == y[1]) & (x[2] == y[2]) & (x[3] == y[3]);
    }
    same = __syncthreads_and(same);
    if (threadIdx.x == 0) eq[r] = same;
}
__global__ void ctx_tail_scan_kernel(const int* __restrict__ eq, int Lc, int* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int s = Lc - 1;                                  // rows s .. Lc-1 are identical to the last row
    while (s > 0 && eq[s - 1]) --s;
    out[0] = s + 1;                                  // effective key count: keys 0 .. s
    out[1] = Lc - s;                                 // key s stands for this many identical keys
}
// tail: int[2 + Lc] (the summary, then the per-row scratch)
static svi_status stage_ctx_tail(const bf16* context, int Lc, int text_dim, int* tail, hipStream_t st) {
    hipLaunchKernelGGL(ctx_tail_rows_kernel, dim3(Lc), dim3(256), 0, st, context, Lc, text_dim, tail + 2);
    hipLaunchKernelGGL(ctx_tail_scan_kernel, dim3(1), dim3(64), 0, st, tail + 2, Lc, tail);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// token -> its 64 (cos, sin) pairs, gathered once from the three axis tables: the q | k normalisation then reads 32 contiguous bytes per
// lane and chunk instead of four 8-byte table lookups behind three integer divisions
__global__ void rope_token_table_kernel(SviRope r, float2* __restrict__ out, int tokens) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)tokens * 64) return;
    const int tok = (int)(i >> 6), pi = (int)(i & 63);
    const int hw = r.h * r.w, pf = tok / hw, rem = tok - pf * hw, ph = rem / r.w, pw = rem - ph * r.w;
    float2 cs;
    if (pi < r.npf) cs = r.tab_f[pf * r.npf + pi];
    else if (pi < r.npf + r.nph) cs = r.tab_h[ph * r.nph + (pi - r.npf)];
    else cs = r.tab_w[pw * r.npw + (pi - r.npf - r.nph)];
    out[i] = cs;
}

// precompute_freqs_cis_3d (models/wan_video_dit.py:161-175): fp64 angles, stored as fp32 (cos, sin)
static svi_status ensure_rope(svi_dit* h, int f, int hh, int ww) {
    if (h->rope_dev && h->rf == f && h->rh == hh && h->rw == ww) return SVI_OK;
    const int dh = 128, d_hw = dh / 3, d_f = dh - 2 * d_hw;
    const int npf = d_f / 2, nph = d_hw / 2, npw = d_hw / 2;
    std::vector<float2> host((size_t)f * npf + (size_t)hh * nph + (size_t)ww * npw);
    size_t o = 0;
    auto fill = [&](int len, int axis_dim, int np) {
        for (int p = 0; p < len; ++p)
            for (int i = 0; i < np; ++i) {
                const double inv = 1.0 / pow(10000.0, (double)(2 * i) / (double)axis_dim);
                const double ang = (double)p * inv;
                host[o++] = make_float2((float)cos(ang), (float)sin(ang));
            }
    };
    fill(f, d_f, npf); fill(hh, d_hw, nph); fill(ww, d_hw, npw);
    if (h->rope_dev) { SVI_CHECK_HIP(hipDeviceSynchronize()); SVI_CHECK_HIP(hipFree(h->rope_dev)); h->rope_dev = nullptr; ++h->generation; }
    const size_t n_axis = (host.size() + 1) & ~(size_t)1;            // the per-token table behind the axis tables, 16-byte aligned
    const size_t tokens = (size_t)f * hh * ww;
    SVI_REQUIRE(tokens * 64 < ((size_t)1 << 31), "rope: grid %dx%dx%d too large", f, hh, ww);
    SVI_CHECK_HIP(hipMalloc((void**)&h->rope_dev, (n_axis + tokens * 64) * sizeof(float2)));
    SVI_CHECK_HIP(hipMemcpy(h->rope_dev, host.data(), host.size() * sizeof(float2), hipMemcpyHostToDevice));
    h->rf = f; h->rh = hh; h->rw = ww;
    h->rope.tab_f = h->rope_dev;
    h->rope.tab_h = h->rope_dev + (size_t)f * npf;
    h->rope.tab_w = h->rope.tab_h + (size_t)hh * nph;
    h->rope.npf = npf; h->rope.nph = nph; h->rope.npw = npw;
    h->rope.f = f; h->rope.h = hh; h->rope.w = ww;
    h->rope.tab_tok = nullptr;
    hipLaunchKernelGGL(rope_token_table_kernel, dim3((unsigned)((tokens * 64 + 255) / 256)), dim3(256), 0, 0, h->rope, h->rope_dev + n_axis, (int)tokens);
    SVI_LAUNCH_CHECK();
    SVI_CHECK_HIP(hipDeviceSynchronize());
    h->rope.tab_tok = h->rope_dev + n_axis;
    return SVI_OK;
}

// ------------------------------------------------------------------------------------------------
// nb > 1: the M rows are nb samples stacked one under the other; the kernel is chosen as for ONE sample (M / nb rows), so a stacked
// launch computes every row with the kernel — and the bits — a per-sample launch would.
static svi_status linear(const bf16* A, int lda, const Lin& l, bf16* C, int ldc, int M, int N, int K, int epi,
                         hipStream_t st, const float* gate = nullptr, const bf16* res = nullptr, int ldres = 0, int nb = 1) {
    SviGemmArgs g{};
    g.A = A; g.lda = lda; g.W = l.w; g.ldw = K; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = l.b; g.epi = epi; g.gate = gate; g.res = res; g.ldres = ldres;
    g.sel_m = nb > 1 ? M / nb : 0;
    return svi_launch_gemm(g, st);
}
// V^T[D, n_tok] = Wv · X^T + bv (bias along rows): same GEMM with the operands swapped.
static svi_status linear_transposed(const bf16* Xin, int ldx, const Lin& l, bf16* CT, int ldct, int n_tok, int N, int K,
                                    hipStream_t st, int nb = 1) {
    SviGemmArgs g{};
    g.A = l.w; g.lda = K; g.W = Xin; g.ldw = ldx; g.C = CT; g.ldc = ldct; g.M = N; g.N = n_tok; g.K = K;
    g.bias = l.b; g.bias_along_m = 1; g.epi = SVI_EPI_BIAS;
    g.sel_n = nb > 1 ? n_tok / nb : 0;
    return svi_launch_gemm(g, st);
}

// Opt-in MX-fp8 projections (svi_dit_proj_mx8): y = epi(quantised(A) · W8^T) for one of the block's square projections.  `A` [R, D] bf16 is quantised into the
// workspace's e4m3 rows + block scales (when `quantise`; else they still hold the rows a previous call of the same phase quantised), W8 = the stored e4m3 weight
// [N = D, K = D] with unit scales.  Sequence-parallel shards keep the bf16 kernels (their launches are shaped for the exchange).
static bool proj_mx8_on(const svi_dit* h) { return h->proj_mx8 && !h->sp_active; }
static svi_status linear_mx8(svi_dit* h, const bf16* A, bool quantise, const unsigned char* W8, const bf16* bias, bf16* C, int ldc, int R, int epi, hipStream_t st,
                             const float* gate = nullptr, const bf16* res = nullptr, int ldres = 0, float* rowss = nullptr, int ldss = 0) {
    Workspace& w = h->ws;
    const int D = h->cfg.dim;
    if (quantise) SVI_TRY(svi_launch_mx8_quantize(A, D, R, D, w.Q8, D, w.S8, w.sc_rows, st));
    SviGemmArgs g{};
    g.A = reinterpret_cast<const bf16*>(w.Q8); g.lda = D; g.W = reinterpret_cast<const bf16*>(W8); g.ldw = D; g.C = C; g.ldc = ldc; g.M = R; g.N = D; g.K = D;
    g.bias = bias; g.epi = epi; g.gate = gate; g.res = res; g.ldres = ldres; g.rowss = rowss; g.ldss = ldss;
    return svi_launch_gemm_mx8(g, w.S8, w.sc_rows, st);
}

// The self-attention third of a block depends on (x, t) only — not on the prompt.  It is written in three pieces so that a
// sequence-parallel shard can exchange heads for tokens around the attention (svi_dit_sp_*): (1) q | k (RMSNorm + RoPE applied,
// q pre-scaled) and V^T of the shard's rows [row0, row0 + L), (2) attention, (3) output projection + gate + residual.
// part: 0 = everything; 1 = LN + modulate and the V^T projection only; 2 = the q | k projection and RMSNorm + RoPE only (after a part-1 call on the
// same rows: the LN output is still in the workspace) — the sequence-parallel gather mode sends V^T on its way while q | k are still being made.
// q8 (opt-in fp8 QK^T attention, plain forward only): RMSNorm + RoPE writes q | k as MX e4m3 rows + block scales into *q8 instead of the bf16 rows of QK.
static svi_status block_qkv(svi_dit* h, int layer, const bf16* X, const float* modf, int L, int row0, bf16* QK, bf16* VT, int ldvt,
                            hipStream_t st, const SviScatter* scatter = nullptr, int nb = 1, int part = 0, const SviQk8* q8 = nullptr) {
    // L = rows of this launch (nb samples of L / nb tokens each, stacked: the two CFG branches of a step on short sequences)
    const svi_dit_config& c = h->cfg;
    const BlockW& b = h->blocks[layer];
    Workspace& w = h->ws;
    const int D = c.dim;
    const float *sh_a = modf, *sc_a = modf + D;
    SviRope rope = h->rope;
    rope.row0 = row0;
    rope.period = nb > 1 ? L / nb : 0;
    // --- self attention: x += gate_msa * o(attn(rope(rms(q)), rope(rms(k)), v))     dit:358,369,226-242
    if (part != 2) { SviProfScope _p(PROF_LN, st); SVI_TRY(svi_launch_ln_mod(X, D, w.Hb, D, L, D, c.eps, nullptr, nullptr, sh_a, sc_a, st)); }
    if (part == 1) { SviProfScope _p(PROF_GEMM_QKV, st); return linear_transposed(w.Hb, D, b.sa.v, VT, ldvt, L, D, D, st, nb); }
    // q | k as ONE N = 2D launch (dit:227-228 side by side): tile columns below D multiply by Wq, the others by Wk — the LN output is read
    // once for both, the weights stay the bound tensors (nothing is packed).  Per element the kernel and the k order are those of the two
    // separate launches (the kernel choice is pinned to the per-projection shape through sel_n): the same bits.
    if (proj_mx8_on(h) && part == 0 && !scatter) {
        // opt-in MX-fp8: the LayerNorm output is quantised ONCE; q and k are two N = D launches on it, V^T the W-scaled form (the stored weight is the A operand)
        SviProfScope _p(PROF_GEMM_QKV, st);
        SVI_TRY(linear_mx8(h, w.Hb, true, b.proj_8[0], b.sa.q.b, QK, 2 * D, L, SVI_EPI_BIAS, st));
        SVI_TRY(linear_mx8(h, w.Hb, false, b.proj_8[1], b.sa.k.b, QK + D, 2 * D, L, SVI_EPI_BIAS, st));
        SviGemmArgs g{};
        g.A = reinterpret_cast<const bf16*>(b.proj_8[2]); g.lda = D; g.W = reinterpret_cast<const bf16*>(w.Q8); g.ldw = D; g.C = VT; g.ldc = ldvt; g.M = D; g.N = L; g.K = D;
        g.bias = b.sa.v.b; g.bias_along_m = 1; g.epi = SVI_EPI_BIAS;
        SVI_TRY(svi_launch_gemm_mx8_wscaled(g, w.S8, w.sc_rows, st));
    } else if (!svi_switches().qk_fused) {
        { SviProfScope _p(PROF_GEMM_QKV, st); SVI_TRY(linear(w.Hb, D, b.sa.q, QK, 2 * D, L, D, D, SVI_EPI_BIAS, st, nullptr, nullptr, 0, nb)); }
        { SviProfScope _p(PROF_GEMM_QKV, st); SVI_TRY(linear(w.Hb, D, b.sa.k, QK + D, 2 * D, L, D, D, SVI_EPI_BIAS, st, nullptr, nullptr, 0, nb)); }
    } else {
      SviProfScope _p(PROF_GEMM_QKV, st);
      SviGemmArgs g{};
      g.A = w.Hb; g.lda = D; g.W = b.sa.q.w; g.ldw = D; g.C = QK; g.ldc = 2 * D; g.M = L; g.N = 2 * D; g.K = D;
      g.bias = b.sa.q.b; g.epi = SVI_EPI_BIAS;
      g.W2 = b.sa.k.w; g.bias2 = b.sa.k.b; g.n_split = D;
      g.sel_m = nb > 1 ? L / nb : 0; g.sel_n = D;
      SVI_TRY(svi_launch_gemm(g, st)); }
    if (part == 0 && !(proj_mx8_on(h) && !scatter)) { SviProfScope _p(PROF_GEMM_QKV, st); SVI_TRY(linear_transposed(w.Hb, D, b.sa.v, VT, ldvt, L, D, D, st, nb)); }
    // q and k in one launch (grid.y = operand): q additionally carries softmax_scale * log2(e) into its single final rounding
    if (q8) { SviProfScope _p(PROF_RMS_ROPE, st); return svi_launch_rmsnorm_rope2_q8(QK, 2 * D, L, D, b.sa.norm_q, b.sa.norm_k, c.eps, &rope, SVI_QK_SCALE_LOG2E, 1.0f, st, *q8); }
    { SviProfScope _p(PROF_RMS_ROPE, st); SVI_TRY(svi_launch_rmsnorm_rope2(QK, 2 * D, L, D, b.sa.norm_q, b.sa.norm_k, c.eps, &rope, SVI_QK_SCALE_LOG2E, 1.0f, st, scatter)); }
    return SVI_OK;
}

static svi_status block_attn_out(svi_dit* h, int layer, bf16* X, const bf16* attn, const float* modf, int L, hipStream_t st, int nb = 1) {
    const svi_dit_config& c = h->cfg;
    const BlockW& b = h->blocks[layer];
    const int D = c.dim;
    const float* g_a = modf + 2 * D;
    if (proj_mx8_on(h)) { SviProfScope _p(PROF_GEMM_O, st); return linear_mx8(h, attn, true, b.proj_8[3], b.sa.o.b, X, D, L, SVI_EPI_BIAS_GATE_RES, st, g_a, X, D); }
    { SviProfScope _p(PROF_GEMM_O, st); SVI_TRY(linear(attn, D, b.sa.o, X, D, L, D, D, SVI_EPI_BIAS_GATE_RES, st, g_a, X, D, nb)); }
    return SVI_OK;
}

// L = tokens of ONE sample; X holds nb samples stacked (nb * L rows).
static svi_status run_block_self(svi_dit* h, int layer, bf16* X, const float* modf, int L, hipStream_t st, int nb = 1) {
    Workspace& w = h->ws;
    const int D = h->cfg.dim;
    // opt-in fp8 QK^T: where the attention will take that kernel, RMSNorm + RoPE writes its operands (e4m3 rows + block scales) instead of bf16 q | k
    SviQk8 q8{};
    bool fused = false;
    if (svi_switches().qk8_fused && svi_rmsnorm_rope_q8_ok(D, &h->rope)) SVI_TRY(svi_flash_qk8_prepare(L, L, h->cfg.num_heads, st, &q8, &fused, nb));
    SVI_TRY(block_qkv(h, layer, X, modf, nb * L, 0, w.QK, w.VT, w.ldvt, st, nullptr, nb, 0, fused ? &q8 : nullptr));
    for (int s = 0; s < nb; ++s) {          // sample s attends over its own token rows / V^T columns [s L, (s+1) L)
        const size_t ro = (size_t)s * L;
        SviProfScope _p(PROF_FLASH_SELF, st);
        const SviQk8 q8s = svi_qk8_sample(q8, s, L, L);
        SVI_TRY(svi_launch_flash(w.QK + ro * 2 * D, 2 * D, w.QK + ro * 2 * D + D, 2 * D, w.VT + ro, w.ldvt, w.Hb + ro * D, D, L, L, h->cfg.num_heads, 1, st, nullptr, fused ? &q8s : nullptr));
    }
    return block_attn_out(h, layer, X, w.Hb, modf, nb * L, st, nb);
}

// Block `layer`'s cross-attention K / V^T of one projected context (dit:272-274; with the CLIP branch also k_img / v_img, :289-291): what the
// context cache keeps per entry.  CTX = [257 CLIP rows (has_image_input) | Lc text rows] x dim.
static svi_status fill_block_kv(svi_dit* h, int layer, const bf16* CTX, int Lc, const CtxKV& kv, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    const BlockW& b = h->blocks[layer];
    Workspace& w = h->ws;
    const int D = c.dim;
    const int img = c.has_image_input ? 257 : 0;
    const bf16* ctx_txt = CTX + (size_t)img * D;
    { SviProfScope _p(PROF_GEMM_CROSS, st); SVI_TRY(linear(ctx_txt, D, b.ca.k, kv.CK, D, Lc, D, D, SVI_EPI_BIAS, st)); }
    { SviProfScope _p(PROF_RMS_ROPE, st); SVI_TRY(svi_launch_rmsnorm_rope(kv.CK, D, Lc, D, b.ca.norm_k, c.eps, nullptr, 1.0f, st)); }
    { SviProfScope _p(PROF_GEMM_CROSS, st); SVI_TRY(linear_transposed(ctx_txt, D, b.ca.v, kv.CVT, w.ldcvt, Lc, D, D, st)); }
    if (img) {
        SVI_TRY(linear(CTX, D, b.ca.k_img, kv.CKi, D, img, D, D, SVI_EPI_BIAS, st));
        SVI_TRY(svi_launch_rmsnorm_rope(kv.CKi, D, img, D, b.ca.norm_k_img, c.eps, nullptr, 1.0f, st));
        SVI_TRY(linear_transposed(CTX, D, b.ca.v_img, kv.CVTi, w.ldcvti, img, D, D, st));
    }
    return SVI_OK;
}

// Cross-attention and MLP thirds of a block.
// nb > 1: X holds nb samples stacked (nb * L rows); sample s attends to its own context (CTXs[s], kvs[s]) — the conditional and the
// unconditional prompt of a CFG step.  Row-local work (norms, projections, MLP) runs once over all rows.
static svi_status block_audio(svi_dit* h, int layer, bf16* X, int L, int f, hipStream_t st);
static svi_status run_block_rest_n(svi_dit* h, int layer, bf16* X, const bf16* const* CTXs, const float* modf, int L, int Lc,
                                   const CtxKV* kvs, int nb, hipStream_t st, int audio_frames = 0) {
    const svi_dit_config& c = h->cfg;
    const BlockW& b = h->blocks[layer];
    Workspace& w = h->ws;
    const int D = c.dim, F = c.ffn_dim, H = c.num_heads;
    const int img = c.has_image_input ? 257 : 0;
    const int R = nb * L;
    const float *sh_m = modf + 3 * D, *sc_m = modf + 4 * D, *g_m = modf + 5 * D;
    // --- cross attention: x += o(attn(rms(q(norm3 x)), rms(k ctx), v ctx) [+ image branch])   dit:370,266-303
    { SviProfScope _p(PROF_LN, st); SVI_TRY(svi_launch_ln_mod(X, D, w.Hb, D, R, D, c.eps, b.norm3_w, b.norm3_b, nullptr, nullptr, st)); }
    // The query's RMSNorm (full width, dit:296) is applied by the attention kernel as it reads q: the projection's epilogue leaves the row sums of squares
    // (per 64-column group; one tiny kernel folds them into rs[row]), so the normalised q never makes its own trip through HBM.  SVI_CROSS_FUSED=0:
    // normalise in place, then attend (rounds 1-4).  The two differ only where the order of the fp32 sum of squares moves a bf16 rounding of q.
    const bool fused = svi_switches().cross_fused && D % 64 == 0 && D % 8 == 0 && R <= w.ldss;
    if (proj_mx8_on(h)) {
        SviProfScope _p(PROF_GEMM_CROSS, st);
        SVI_TRY(linear_mx8(h, w.Hb, true, b.proj_8[4], b.ca.q.b, w.QK, 2 * D, R, SVI_EPI_BIAS, st, nullptr, nullptr, 0, fused ? w.RSS : nullptr, fused ? w.ldss : 0));
    } else {
        SviProfScope _p(PROF_GEMM_CROSS, st);
        SviGemmArgs g{};
        g.A = w.Hb; g.lda = D; g.W = b.ca.q.w; g.ldw = D; g.C = w.QK; g.ldc = 2 * D; g.M = R; g.N = D; g.K = D;
        g.bias = b.ca.q.b; g.epi = SVI_EPI_BIAS;
        g.sel_m = nb > 1 ? R / nb : 0;
        if (fused) { g.rowss = w.RSS; g.ldss = w.ldss; }
        SVI_TRY(svi_launch_gemm(g, st));
    }
    if (fused) { SviProfScope _p(PROF_FLASH_CROSS, st); SVI_TRY(svi_launch_row_rs(w.RSS, D / 64, w.ldss, R, D, c.eps, w.RS, st)); }      // (timed with the attention it feeds)
    else { SviProfScope _p(PROF_RMS_ROPE, st); SVI_TRY(svi_launch_rmsnorm_rope(w.QK, 2 * D, R, D, b.ca.norm_q, c.eps, nullptr, SVI_QK_SCALE_LOG2E, st)); }
    for (int s = 0; s < nb; ++s) {
        const CtxKV& kv = kvs[s];
        const bf16* CTX = CTXs[s];
        const size_t ro = (size_t)s * L;
        const SviQNorm qn{w.RS + ro, b.ca.norm_q, SVI_QK_SCALE_LOG2E};
        if (kv.compute) SVI_TRY(fill_block_kv(h, layer, CTX, Lc, kv, st));
        {
            SviProfScope _p(PROF_FLASH_CROSS, st);
            if (fused) SVI_TRY(svi_launch_flash_cross(w.QK + ro * 2 * D, 2 * D, kv.CK, D, kv.CVT, w.ldcvt, w.Hb + ro * D, D, L, Lc, H, st, kv.tail, &qn, kv.key_blocks));
            else SVI_TRY(svi_launch_flash(w.QK + ro * 2 * D, 2 * D, kv.CK, D, kv.CVT, w.ldcvt, w.Hb + ro * D, D, L, Lc, H, 1, st, kv.tail));
        }
        if (img) {
            if (fused) SVI_TRY(svi_launch_flash_cross(w.QK + ro * 2 * D, 2 * D, kv.CKi, D, kv.CVTi, w.ldcvti, w.A2 + ro * D, D, L, img, H, st, nullptr, &qn));
            else SVI_TRY(svi_launch_flash(w.QK + ro * 2 * D, 2 * D, kv.CKi, D, kv.CVTi, w.ldcvti, w.A2 + ro * D, D, L, img, H, 1, st));
        }
    }
    if (img) SVI_TRY(svi_launch_add_bf16(w.Hb, w.A2, (int64_t)R * D, st));
    if (proj_mx8_on(h)) { SviProfScope _p(PROF_GEMM_CROSS, st); SVI_TRY(linear_mx8(h, w.Hb, true, b.proj_8[5], b.ca.o.b, X, D, R, SVI_EPI_BIAS_GATE_RES, st, nullptr, X, D)); }
    else { SviProfScope _p(PROF_GEMM_CROSS, st); SVI_TRY(linear(w.Hb, D, b.ca.o, X, D, R, D, D, SVI_EPI_BIAS_GATE_RES, st, nullptr, X, D, nb)); }
    // --- talk variant: audio cross-attention between the text cross-attention and the MLP (dit:361-366)
    if (audio_frames > 0) SVI_TRY(block_audio(h, layer, X, L, audio_frames, st));
    // --- MLP: x += gate_mlp * W2 gelu_tanh(W1 modulate(norm2 x))                  dit:372-373,334-335
    { SviProfScope _p(PROF_LN, st); SVI_TRY(svi_launch_ln_mod(X, D, w.Hb, D, R, D, c.eps, nullptr, nullptr, sh_m, sc_m, st)); }
    if (h->ffn_mx8) {          // opt-in MX-fp8 MLP: both GEMMs on v_mfma_scale_f32_32x32x64_f8f6f4, activations quantised per 32-element K block
        auto mx = [&](const unsigned char* A8, int K, const unsigned char* W8, const bf16* bias, bf16* C, int ldc, int N, int epi, const float* gate, const bf16* res) {
            SviGemmArgs g{};
            g.A = reinterpret_cast<const bf16*>(A8); g.lda = K; g.W = reinterpret_cast<const bf16*>(W8); g.ldw = K; g.C = C; g.ldc = ldc; g.M = R; g.N = N; g.K = K;
            g.bias = bias; g.epi = epi; g.gate = gate; g.res = res; g.ldres = ldc;
            return svi_launch_gemm_mx8(g, w.S8, w.sc_rows, st);
        };
        if (svi_switches().mx8_fused && F % 256 == 0 && (size_t)(F / 128) * w.sc_rows * 4 + 256 <= (size_t)R * F) {
            // ffn1 quantises its own GELU output in the epilogue: the [R, F] bf16 activation is never written or read back.  Its e4m3 bytes and block
            // scales live where that tensor would have been (w.Fb: 2 R F bytes >= R F + R F / 32)
            unsigned char* q8f = reinterpret_cast<unsigned char*>(w.Fb);
            unsigned* s8f = reinterpret_cast<unsigned*>(q8f + (((size_t)R * F + 255) & ~(size_t)255));
            { SviProfScope _p(PROF_GEMM_FFN1, st);
              SVI_TRY(svi_launch_mx8_quantize(w.Hb, D, R, D, w.Q8, D, w.S8, w.sc_rows, st));
              SviGemmArgs g{};
              g.A = reinterpret_cast<const bf16*>(w.Q8); g.lda = D; g.W = reinterpret_cast<const bf16*>(b.ffn0_8); g.ldw = D; g.C = w.Fb; g.ldc = F; g.M = R; g.N = F; g.K = D;
              g.bias = b.ffn0.b; g.epi = SVI_EPI_BIAS_GELU_TANH;
              g.q8 = q8f; g.ldq8 = F; g.q8s = s8f; g.q8_sc_rows = w.sc_rows;
              SVI_TRY(svi_launch_gemm_mx8(g, w.S8, w.sc_rows, st)); }
            { SviProfScope _p(PROF_GEMM_FFN2, st);
              SviGemmArgs g{};
              g.A = reinterpret_cast<const bf16*>(q8f); g.lda = F; g.W = reinterpret_cast<const bf16*>(b.ffn2_8); g.ldw = F; g.C = X; g.ldc = D; g.M = R; g.N = D; g.K = F;
              g.bias = b.ffn2.b; g.epi = SVI_EPI_BIAS_GATE_RES; g.gate = g_m; g.res = X; g.ldres = D;
              SVI_TRY(svi_launch_gemm_mx8(g, s8f, w.sc_rows, st)); }
            return SVI_OK;
        }
        { SviProfScope _p(PROF_GEMM_FFN1, st);
          SVI_TRY(svi_launch_mx8_quantize(w.Hb, D, R, D, w.Q8, D, w.S8, w.sc_rows, st));
          SVI_TRY(mx(w.Q8, D, b.ffn0_8, b.ffn0.b, w.Fb, F, F, SVI_EPI_BIAS_GELU_TANH, nullptr, nullptr)); }
        { SviProfScope _p(PROF_GEMM_FFN2, st);
          SVI_TRY(svi_launch_mx8_quantize(w.Fb, F, R, F, w.Q8, F, w.S8, w.sc_rows, st));
          SVI_TRY(mx(w.Q8, F, b.ffn2_8, b.ffn2.b, X, D, D, SVI_EPI_BIAS_GATE_RES, g_m, X)); }
        return SVI_OK;
    }
    { SviProfScope _p(PROF_GEMM_FFN1, st); SVI_TRY(linear(w.Hb, D, b.ffn0, w.Fb, F, R, F, D, SVI_EPI_BIAS_GELU_TANH, st, nullptr, nullptr, 0, nb)); }
    { SviProfScope _p(PROF_GEMM_FFN2, st); SVI_TRY(linear(w.Fb, F, b.ffn2, X, D, R, D, F, SVI_EPI_BIAS_GATE_RES, st, g_m, X, D, nb)); }
    return SVI_OK;
}
static svi_status run_block_rest(svi_dit* h, int layer, bf16* X, const bf16* CTX, const float* modf, int L, int Lc,
                                 const CtxKV& kv, hipStream_t st, int audio_frames = 0) {
    return run_block_rest_n(h, layer, X, &CTX, modf, L, Lc, &kv, 1, st, audio_frames);
}

static svi_status run_block(svi_dit* h, int layer, bf16* X, const bf16* CTX, const float* modf, int L, int Lc,
                            const CtxKV& kv, hipStream_t st, int audio_frames = 0) {
    SVI_TRY(run_block_self(h, layer, X, modf, L, st));
    return run_block_rest(h, layer, X, CTX, modf, L, Lc, kv, st, audio_frames);
}

// ---- talk variant ---------------------------------------------------------------------------------------------------------------
#define SVI_AUD_TOK 32          // context tokens per latent frame (AudioProjModel context_tokens, dit:459)
#define SVI_AUD_DIM 768         // their width = encoder_hidden_states_dim of the audio cross-attention (dit:342)
static svi_status ensure_audio(svi_dit* h, int f) {
    if (h->aud_base && h->aud_frames == f) return SVI_OK;
    const size_t D = h->cfg.dim, na = (size_t)f * SVI_AUD_TOK;
    const int ldavt = (int)((na + 7) / 8 * 8);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t oh0 = take((size_t)f * 512 * 2), oh1 = take((size_t)f * 512 * 2), op3 = take(na * SVI_AUD_DIM * 2), oaud = take(na * SVI_AUD_DIM * 2);
    const size_t oak = take(na * D * 2), oavt = take(D * ldavt * 2);
    if (h->aud_base) { SVI_CHECK_HIP(hipFree(h->aud_base)); h->aud_base = nullptr; }
    hipError_t e = hipMalloc((void**)&h->aud_base, off);
    if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B audio workspace) failed: %s", off, hipGetErrorString(e)); return SVI_ERR_OOM; }
    SVI_CHECK_HIP(hipMemset(h->aud_base, 0, off));
    SVI_CHECK_HIP(hipDeviceSynchronize());
    auto P = [&](size_t o) { return reinterpret_cast<bf16*>(h->aud_base + o); };
    h->AH0 = P(oh0); h->AH1 = P(oh1); h->AP3 = P(op3); h->AUD = P(oaud); h->AK = P(oak); h->AVT = P(oavt);
    h->ldavt = ldavt; h->aud_frames = f;
    return SVI_OK;
}
// audio_embed = audio_proj(first, latter) (AudioProjModel.forward, dit:82-115) -> AUD bf16 [f * 32, 768]: 32 context tokens per frame
static svi_status stage_audio(svi_dit* h, int f, hipStream_t st) {
    SVI_REQUIRE(h->cfg.enable_multitalk, "audio was set on a model without enable_multitalk");
    SVI_REQUIRE(h->aud_latter_n == f - 1, "audio windows cover %d latent frames, the latents have %d", h->aud_latter_n + 1, f);
    SVI_TRY(ensure_audio(h, f));
    const int k0 = 5 * 12 * SVI_AUD_DIM, k1 = 8 * 12 * SVI_AUD_DIM;
    SVI_TRY(linear(h->aud_first, k0, h->ap1, h->AH0, 512, 1, 512, k0, SVI_EPI_BIAS_RELU, st));                       // relu(proj1), frame 0
    if (f > 1) SVI_TRY(linear(h->aud_latter, k1, h->ap1vf, h->AH0 + 512, 512, f - 1, 512, k1, SVI_EPI_BIAS_RELU, st));   // relu(proj1_vf), frames 1..
    SVI_TRY(linear(h->AH0, 512, h->ap2, h->AH1, 512, f, 512, 512, SVI_EPI_BIAS_RELU, st));                          // relu(proj2)
    SVI_TRY(linear(h->AH1, 512, h->ap3, h->AP3, SVI_AUD_TOK * SVI_AUD_DIM, f, SVI_AUD_TOK * SVI_AUD_DIM, 512, SVI_EPI_BIAS, st));   // proj3 -> [f, 32*768]
    // nn.LayerNorm(768) (eps 1e-5) over every context token
    return svi_launch_ln_mod(h->AP3, SVI_AUD_DIM, h->AUD, SVI_AUD_DIM, f * SVI_AUD_TOK, SVI_AUD_DIM, 1e-5f, h->ap_norm_w, h->ap_norm_b, nullptr, nullptr, st);
}
// x += proj(attention_per_frame(q_linear(norm_x(x)), kv_linear(audio)))   (DiTBlock.forward dit:361-366; SingleStreamAttention.forward,
// models/attention.py:318-371 with human_num == 1): frame fr's h*w tokens attend to that frame's 32 audio tokens, scale head_dim^-0.5,
// no q/k norm, no RoPE.  K | V of the audio tokens depend on (audio, weights) only; they are re-projected per forward (3 GFLOP per block).
static svi_status block_audio(svi_dit* h, int layer, bf16* X, int L, int f, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    const BlockW& b = h->blocks[layer];
    Workspace& w = h->ws;
    const int D = c.dim, na = f * SVI_AUD_TOK, S = L / f;
    const Lin lk{b.aud_kv.w, b.aud_kv.b}, lv{b.aud_kv.w + (size_t)D * SVI_AUD_DIM, b.aud_kv.b + D};      // kv_linear rows [0, D) = K, [D, 2D) = V
    SVI_TRY(svi_launch_ln_mod(X, D, w.Hb, D, L, D, c.eps, b.normx_w, b.normx_b, nullptr, nullptr, st));
    SVI_TRY(linear(w.Hb, D, b.aud_q, w.QK, 2 * D, L, D, D, SVI_EPI_BIAS, st));
    SVI_TRY(linear(h->AUD, SVI_AUD_DIM, lk, h->AK, D, na, D, SVI_AUD_DIM, SVI_EPI_BIAS, st));
    SVI_TRY(linear_transposed(h->AUD, SVI_AUD_DIM, lv, h->AVT, h->ldavt, na, D, SVI_AUD_DIM, st));
    for (int fr = 0; fr < f; ++fr)
        SVI_TRY(svi_launch_flash(w.QK + (size_t)fr * S * 2 * D, 2 * D, h->AK + (size_t)fr * SVI_AUD_TOK * D, D, h->AVT + (size_t)fr * SVI_AUD_TOK, h->ldavt,
                                 w.Hb + (size_t)fr * S * D, D, S, SVI_AUD_TOK, c.num_heads, 0, st));
    return linear(w.Hb, D, b.aud_proj, X, D, L, D, D, SVI_EPI_BIAS_GATE_RES, st, nullptr, X, D);
}

// modf[i][c] = bf16(modulation[i][c] + t_mod[i][c]); rows in scale_mask store bf16(1 + that)
// (DiTBlock.forward models/wan_video_dit.py:356-357, modulate :150-151, Head.forward :402)
__global__ void mod_prepare_one_kernel(const bf16* __restrict__ modulation, const bf16* __restrict__ tmod,
                                       float* __restrict__ modf, int D, int rows, int scale_mask, int tmod_rows) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * D) return;
    const int i = idx / D, c = idx - i * D;
    const float t = (float)tmod[(tmod_rows == 1 ? 0 : i) * D + c];
    float m = rbf((float)modulation[idx] + t);
    if ((scale_mask >> i) & 1) m = rbf(1.0f + m);
    modf[idx] = m;
}

static svi_status mod_one(const bf16* modulation, const bf16* tmod, float* modf, int D, int rows, int scale_mask,
                          int tmod_rows, hipStream_t st) {
    const int n = rows * D;
    hipLaunchKernelGGL(mod_prepare_one_kernel, dim3((n + 255) / 256), dim3(256), 0, st, modulation, tmod, modf, D, rows,
                       scale_mask, tmod_rows);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ---- stages of one forward (svi_video.py:74-137).  forward_one() runs them in the reference's order; forward_pair() shares
// the stages that do not depend on the prompt between the conditional and the unconditional forward of a CFG step.
struct CtxUse { CtxEntry* ce; bf16* CTXp; bool compute; int key_blocks; };

// timestep embedding -> t, t_mod, per-block modulation rows                     svi_video.py:92-93
static svi_status stage_time(svi_dit* h, const float* timestep, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    const int D = c.dim;
    Workspace& w = h->ws;
    hipLaunchKernelGGL(sinusoid_kernel, dim3((c.freq_dim / 2 + 63) / 64), dim3(64), 0, st, timestep, w.e, c.freq_dim);
    SVI_LAUNCH_CHECK();
    SVI_TRY(linear(w.e, c.freq_dim, h->time0, w.h1, D, 1, D, c.freq_dim, SVI_EPI_BIAS_SILU, st));
    SVI_TRY(linear(w.h1, D, h->time2, w.t, D, 1, D, D, SVI_EPI_BIAS, st));
    hipLaunchKernelGGL(silu_kernel, dim3((D + 255) / 256), dim3(256), 0, st, w.t, w.st, D);
    SVI_LAUNCH_CHECK();
    SVI_TRY(linear(w.st, D, h->timeproj, w.tmod, 6 * D, 1, 6 * D, D, SVI_EPI_BIAS, st));
    for (int l = 0; l < c.num_layers; ++l)
        SVI_TRY(mod_one(h->blocks[l].modulation, w.tmod, w.modf + (size_t)l * 6 * D, D, 6, (1 << 1) | (1 << 4), 6, st));
    SVI_TRY(mod_one(h->head_mod, w.t, w.headf, D, 2, 1 << 1, 1, st));
    return SVI_OK;
}

// text_embedding(context) (and img_emb(clip_feature) in front of it) -> CTXp [257 | Lc rows, dim]; the identical-suffix summary -> tail.   svi_video.py:94-99
static svi_status project_context(svi_dit* h, const bf16* context, const bf16* clip, int Lc, bf16* CTXp, int* tail, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    const int D = c.dim;
    Workspace& w = h->ws;
    const int img = c.has_image_input ? 257 : 0;
    SVI_TRY(stage_ctx_tail(context, Lc, c.text_dim, tail, st));
    SVI_TRY(linear(context, c.text_dim, h->text0, w.CTXH, D, Lc, D, c.text_dim, SVI_EPI_BIAS_GELU_TANH, st));
    SVI_TRY(linear(w.CTXH, D, h->text2, CTXp + (size_t)img * D, D, Lc, D, D, SVI_EPI_BIAS, st));
    if (img) {
        SVI_REQUIRE(clip, "has_image_input model needs clip_feature");
        SVI_TRY(svi_launch_ln_mod(clip, 1280, w.IMG0, 1280, 257, 1280, 1e-5f, h->img_ln0_w, h->img_ln0_b, nullptr, nullptr, st));
        SVI_TRY(linear(w.IMG0, 1280, h->img1, w.IMG1, 1280, 257, 1280, 1280, SVI_EPI_BIAS_GELU_ERF, st));
        SVI_TRY(linear(w.IMG1, 1280, h->img3, w.IMGD, D, 257, D, 1280, SVI_EPI_BIAS, st));      // own [264, D] rows: Hb holds only L rows (L < 257 on tiny grids / shards)
        SVI_TRY(svi_launch_ln_mod(w.IMGD, D, CTXp, D, 257, D, 1e-5f, h->img_ln4_w, h->img_ln4_b, nullptr, nullptr, st));
    }
    return SVI_OK;
}

// How many keys the cross-attention will walk for this prompt decides its kernel (svi_launch_flash_cross: up to 128 keys stay resident in LDS, in a kernel
// instantiated per 32-key block count); the count is made on the device (ctx_tail_scan_kernel).  It is read back once per projection — here, outside any
// capture — so that the launches need no device-side dispatch.  Under capture (or on any error) it stays unknown (0: the streaming kernel, any count).
static int read_key_blocks(const int* tail_dev, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    int n = 0;
    if (hipMemcpyAsync(&n, tail_dev, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0 ? (n + 31) / 32 : 0;
}

// text (and CLIP image) context: projected here or taken from the context cache      svi_video.py:94-99
static svi_status stage_context(svi_dit* h, const bf16* context, const bf16* clip, const bf16* y, int Lc, CtxUse* use, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    Workspace& w = h->ws;
    const int img = c.has_image_input ? 257 : 0;
    CtxEntry* ce = nullptr;
    bool ctx_compute = true;
    if (h->ctx_cache_on) {
        CtxEntry* lru = &h->ctx_entries[0];
        for (auto& e : h->ctx_entries) {
            if (e.filled && e.key_ctx == (const void*)context && e.key_clip == (const void*)clip && e.Lc == Lc) { ce = &e; break; }
            if (e.stamp < lru->stamp) lru = &e;
        }
        if (ce) ctx_compute = false;
        else {
            ce = lru;
            const size_t Dd = c.dim, Lctx = (size_t)Lc + img, nl = c.num_layers;
            const size_t per_layer = al(Lctx * Dd * 2) + al(Dd * w.ldcvt * 2) + (img ? al((size_t)264 * Dd * 2) + al(Dd * 264 * 2) : 0);
            const size_t need = al(Lctx * Dd * 2) + nl * per_layer + al((size_t)(Lc + 8) * 4);
            if (!ce->base || ce->Lc != Lc || ce->CK.size() != nl) {
                if (ce->base) { SVI_CHECK_HIP(hipFree(ce->base)); ce->base = nullptr; }
                hipError_t e2 = hipMalloc((void**)&ce->base, need);
                if (e2 != hipSuccess) { svi_set_error("hipMalloc(%zu B context cache) failed: %s", need, hipGetErrorString(e2)); return SVI_ERR_OOM; }
                SVI_CHECK_HIP(hipMemsetAsync(ce->base, 0, need, st));      // V^T pad columns must read as zeros
                size_t off = 0;
                auto take = [&](size_t bytes) { bf16* p = reinterpret_cast<bf16*>(ce->base + off); off += al(bytes); return p; };
                ce->CTX = take(Lctx * Dd * 2);
                ce->CK.assign(nl, nullptr); ce->CVT.assign(nl, nullptr); ce->CKi.assign(nl, nullptr); ce->CVTi.assign(nl, nullptr);
                for (size_t l = 0; l < nl; ++l) {
                    ce->CK[l] = take(Lctx * Dd * 2); ce->CVT[l] = take(Dd * w.ldcvt * 2);
                    if (img) { ce->CKi[l] = take((size_t)264 * Dd * 2); ce->CVTi[l] = take(Dd * 264 * 2); }
                }
                ce->tail = reinterpret_cast<int*>(take((size_t)(Lc + 8) * 4));
            }
            ce->key_ctx = context; ce->key_clip = clip; ce->Lc = Lc; ce->filled = true;
            ++h->generation;                                    // an entry was (re)filled: pointers / contents a captured graph relies on moved
        }
        ce->stamp = ++h->ctx_clock;
    }
    bf16* CTXp = ce ? ce->CTX : w.CTX;
    if (img) SVI_REQUIRE(clip && y, "has_image_input model needs clip_feature and y");
    int key_blocks = ce ? ce->key_blocks : 0;
    if (ctx_compute) {
        SVI_TRY(project_context(h, context, clip, Lc, CTXp, ce ? ce->tail : w.tail, st));
        key_blocks = (svi_switches().cross_fused && svi_switches().cross_dedup) ? read_key_blocks(ce ? ce->tail : w.tail, st) : 0;
        if (ce) ce->key_blocks = key_blocks;
    }
    use->ce = ce; use->CTXp = CTXp; use->compute = ctx_compute; use->key_blocks = key_blocks;
    return SVI_OK;
}

// patchify -> X                                                                svi_video.py:101, dit:473-477
static svi_status stage_embed(svi_dit* h, const bf16* x, const bf16* y, const bf16* addc, int T, int H, int W, int L, hipStream_t st,
                              int row0 = 0) {       // row0: first token of a sequence-parallel shard (L = its row count)
    const svi_dit_config& c = h->cfg;
    Workspace& w = h->ws;
    const int64_t n = (int64_t)L * w.kpatch;
    hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, w.PATCH, 16,
                       c.in_dim - 16, T, H, W, c.patch_t, c.patch_h, c.patch_w, w.kpatch, L, row0);
    SVI_LAUNCH_CHECK();
    Lin pe{h->patch_w, h->patch_b};
    SVI_TRY(linear(w.PATCH, w.kpatch, pe, w.X, c.dim, L, c.dim, w.kpatch, SVI_EPI_BIAS, st));
    if (addc) SVI_TRY(svi_launch_add_bf16(w.X, addc + (size_t)row0 * c.dim, (int64_t)L * c.dim, st));
    return SVI_OK;
}

static CtxKV kv_of(svi_dit* h, const CtxUse& u, int l) {
    Workspace& w = h->ws;
    const int* tail = svi_switches().cross_dedup ? (u.ce ? u.ce->tail : w.tail) : nullptr;
    return CtxKV{u.ce ? u.ce->CK[l] : w.CK, u.ce ? u.ce->CVT[l] : w.CVT, u.ce ? u.ce->CKi[l] : w.CKi, u.ce ? u.ce->CVTi[l] : w.CVTi, u.compute, tail,
                 u.key_blocks};
}

// head (dit:401-404): LN + modulation + Linear(dim -> out_dim * patch volume) on the rows in X -> HO [L, ho_ld]
static int head_ld(const svi_dit_config& c) { return (c.out_dim * c.patch_t * c.patch_h * c.patch_w + 7) / 8 * 8; }
static svi_status stage_head_rows(svi_dit* h, bf16* HO, int L, hipStream_t st, int nb = 1) {
    const svi_dit_config& c = h->cfg;
    Workspace& w = h->ws;
    const int D = c.dim;
    const int ho = c.out_dim * c.patch_t * c.patch_h * c.patch_w;
    SVI_TRY(svi_launch_ln_mod(w.X, D, w.Hb, D, L, D, c.eps, nullptr, nullptr, w.headf, w.headf + D, st));
    return linear(w.Hb, D, h->head, HO, head_ld(c), L, ho, D, SVI_EPI_BIAS, st, nullptr, nullptr, 0, nb);
}
// unpatchify (dit:479-484) of all L = f*h*w head rows
static svi_status stage_unpatchify(svi_dit* h, const bf16* HO, bf16* out, int T, int H, int W, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    const int64_t n = (int64_t)c.out_dim * T * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, HO, out, c.out_dim, T,
                       H, W, c.patch_t, c.patch_h, c.patch_w, head_ld(c));
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}
static svi_status stage_head(svi_dit* h, bf16* out, int T, int H, int W, int L, hipStream_t st) {
    SVI_TRY(stage_head_rows(h, h->ws.HO, L, st));
    return stage_unpatchify(h, h->ws.HO, out, T, H, W, st);
}

// tea_mode (TeaCache, pipelines/svi_video.py:23-72,117-131): 0 = plain forward; 1 = run the blocks and leave
// residual = bf16(x_after_blocks - x_before_blocks) in `residual` (TeaCache.store); 2 = skip the blocks, x = bf16(x + residual)
// (TeaCache.update).  The skip decision itself is host logic on t_mod (svi_dit_time_mod + svi_hip.TeaCache).
static svi_status forward_one(svi_dit* h, const bf16* x, const float* timestep, const bf16* context, const bf16* clip,
                              const bf16* y, const bf16* addc, bf16* out, int T, int H, int W, int Lc, hipStream_t st,
                              int tea_mode = 0, bf16* residual = nullptr) {
    const svi_dit_config& c = h->cfg;
    const int D = c.dim;
    const int f = T / c.patch_t, hh = H / c.patch_h, ww = W / c.patch_w;
    const int L = f * hh * ww;
    SVI_TRY(ensure_workspace(h, L, Lc, st));
    SVI_TRY(ensure_rope(h, f, hh, ww));
    Workspace& w = h->ws;
    SVI_TRY(stage_time(h, timestep, st));
    if (tea_mode == 2) {
        SVI_TRY(stage_embed(h, x, y, addc, T, H, W, L, st));
        SVI_TRY(svi_launch_add_bf16(w.X, residual, (int64_t)L * D, st));
        return stage_head(h, out, T, H, W, L, st);
    }
    CtxUse cu{};
    SVI_TRY(stage_context(h, context, clip, y, Lc, &cu, st));
    SVI_TRY(stage_embed(h, x, y, addc, T, H, W, L, st));
    const int audio_frames = h->aud_first ? f : 0;                       // talk variant armed (svi_dit_set_audio)
    if (audio_frames) SVI_TRY(stage_audio(h, f, st));
    if (tea_mode == 1) SVI_CHECK_HIP(hipMemcpyAsync(w.X2, w.X, (size_t)L * D * 2, hipMemcpyDeviceToDevice, st));
    for (int l = 0; l < c.num_layers; ++l)
        SVI_TRY(run_block(h, l, w.X, cu.CTXp, w.modf + (size_t)l * 6 * D, L, Lc, kv_of(h, cu, l), st, audio_frames));
    if (tea_mode == 1) SVI_TRY(svi_launch_sub_bf16(residual, w.X, w.X2, (int64_t)L * D, st));
    return stage_head(h, out, T, H, W, L, st);
}

// The two forwards of a classifier-free-guidance step (svi_video.py:401-408) see the same latents and timestep and differ only
// in the prompt embedding, which first enters in block 0's cross-attention.  Everything before that — timestep embedding and
// the 31 modulation rows, patchify + patch embedding, and block 0's self-attention third (LN, q/k/v, RMSNorm+RoPE, flash, o) —
// is computed once and its X snapshot restored for the second forward: 1/60 of a step's self-attention work, results bit-
// identical to two svi_dit_forward calls (same kernels on the same operands; tests/test_gpu_dit.py).
//
// After the shared part the two branches are STACKED — X holds 2 L rows, conditional on top — and every row-local kernel (norms, projections,
// MLP, head) runs once over both: half the launches, GEMMs that offered 120 tiles to 256 CUs at BASELINE configs[0] (1280 tokens) offer 240, and at
// the C2 size ffn1's 17.5 rounds of tiles per branch become 35 whole ones (same box, alternating: 425.1-428.3 -> 422.2-423.1 ms per step).
// Attention runs per branch (own rows; own prompt K / V).  Taken whenever the widest stacked activation stays below 2 GiB (the GEMM's buffer
// descriptors and row offsets are 32-bit): every BASELINE size at 1.3B and 14B widths; not the 1.3B model at 720p.
// Each GEMM keeps the kernel a one-branch launch would pick (SviGemmArgs.sel_m / sel_n), so outputs stay bit-identical to two
// svi_dit_forward calls.  Needs the context cache (each prompt's projected context and K / V in buffers of its own).
#ifndef SVI_PAIR_STACK_MAX
#define SVI_PAIR_STACK_MAX (1 << 20)       // tokens (A/B builds: -DSVI_PAIR_STACK_MAX=8192 is the round-1..4 behaviour: short sequences only)
#endif
static svi_status forward_pair(svi_dit* h, const bf16* x, const float* timestep, const bf16* ctx_a, const bf16* ctx_b, const bf16* clip,
                               const bf16* y, const bf16* addc, bf16* out_a, bf16* out_b, int T, int H, int W, int Lc, hipStream_t st) {
    const svi_dit_config& c = h->cfg;
    const int D = c.dim;
    const int f = T / c.patch_t, hh = H / c.patch_h, ww = W / c.patch_w;
    const int L = f * hh * ww;
    const size_t widest = (size_t)std::max(c.ffn_dim, 2 * D);
    // The 2 GiB bound covers every buffer the stacked form addresses with 32-bit byte offsets: the widest activation [2 L, max(ffn_dim, 2 dim)] (the GEMMs'
    // buffer descriptors; the q | k buffer is [2 L, 2 dim]) and V^T [dim, ldvt >= 2 L] (the attention kernel's int row offsets: dim * 2 L * 2 bytes, never
    // more than the q | k buffer's).  The MX-fp8 scale tables are smaller than either.
    bool stacked = h->ctx_cache_on && L <= SVI_PAIR_STACK_MAX && L % 8 == 0 && ctx_a != ctx_b && (size_t)2 * L * widest * 2 < ((size_t)1 << 31) &&
                   !(h->pair_stack_oom_rows && 2 * L >= h->pair_stack_oom_rows);
    svi_status ws = ensure_workspace(h, stacked ? 2 * L : L, Lc, st);
    if (ws == SVI_ERR_OOM && stacked) {          // the doubled workspace does not fit (about 5 GB more at the 14B widths): the unstacked form gives the same bits
        (void)hipGetLastError();
        stacked = false;
        if (!h->pair_stack_oom_rows || 2 * L < h->pair_stack_oom_rows) h->pair_stack_oom_rows = 2 * L;      // remembered: see the member's comment
        ws = ensure_workspace(h, L, Lc, st);
    }
    SVI_TRY(ws);
    SVI_TRY(ensure_rope(h, f, hh, ww));
    Workspace& w = h->ws;
    SVI_TRY(stage_time(h, timestep, st));
    SVI_TRY(stage_embed(h, x, y, addc, T, H, W, L, st));
    SVI_TRY(run_block_self(h, 0, w.X, w.modf, L, st));
    const bf16* ctxs[2] = {ctx_a, ctx_b};
    bf16* outs[2] = {out_a, out_b};
    if (stacked) {
        SVI_CHECK_HIP(hipMemcpyAsync(w.X + (size_t)L * D, w.X, (size_t)L * D * 2, hipMemcpyDeviceToDevice, st));
        CtxUse cu[2]{};
        const bf16* CTXs[2];
        for (int k = 0; k < 2; ++k) {
            SVI_TRY(stage_context(h, ctxs[k], clip, y, Lc, &cu[k], st));
            SVI_REQUIRE(cu[k].ce != nullptr, "stacked CFG pair needs the context cache");
            CTXs[k] = cu[k].CTXp;
        }
        for (int l = 0; l < c.num_layers; ++l) {
            const float* modf = w.modf + (size_t)l * 6 * D;
            if (l) SVI_TRY(run_block_self(h, l, w.X, modf, L, st, 2));
            CtxKV kvs[2] = {kv_of(h, cu[0], l), kv_of(h, cu[1], l)};
            SVI_TRY(run_block_rest_n(h, l, w.X, CTXs, modf, L, Lc, kvs, 2, st));
        }
        SVI_TRY(stage_head_rows(h, w.HO, 2 * L, st, 2));
        for (int k = 0; k < 2; ++k) SVI_TRY(stage_unpatchify(h, w.HO + (size_t)k * L * head_ld(c), outs[k], T, H, W, st));
        return SVI_OK;
    }
    SVI_CHECK_HIP(hipMemcpyAsync(w.X2, w.X, (size_t)L * D * 2, hipMemcpyDeviceToDevice, st));
    for (int k = 0; k < 2; ++k) {
        if (k) SVI_CHECK_HIP(hipMemcpyAsync(w.X, w.X2, (size_t)L * D * 2, hipMemcpyDeviceToDevice, st));
        CtxUse cu{};
        SVI_TRY(stage_context(h, ctxs[k], clip, y, Lc, &cu, st));      // the CLIP branch has scratch rows of its own: X is untouched
        SVI_TRY(run_block_rest(h, 0, w.X, cu.CTXp, w.modf, L, Lc, kv_of(h, cu, 0), st));
        for (int l = 1; l < c.num_layers; ++l)
            SVI_TRY(run_block(h, l, w.X, cu.CTXp, w.modf + (size_t)l * 6 * D, L, Lc, kv_of(h, cu, l), st));
        SVI_TRY(stage_head(h, outs[k], T, H, W, L, st));
    }
    return SVI_OK;
}

extern "C" svi_status svi_dit_set_audio(svi_dit* h, const void* audio_first, const void* audio_latter, int32_t n_latter) {
    SVI_REQUIRE(h, "null handle");
    if (!audio_first) { h->aud_first = h->aud_latter = nullptr; h->aud_latter_n = 0; return SVI_OK; }
    SVI_REQUIRE(h->cfg.enable_multitalk, "svi_dit_set_audio: the model was created without enable_multitalk");
    SVI_REQUIRE(n_latter >= 0 && (n_latter == 0 || audio_latter), "svi_dit_set_audio: %d later frames but no audio_latter", n_latter);
    SVI_REQUIRE(((uintptr_t)audio_first % 16) == 0 && ((uintptr_t)audio_latter % 16) == 0, "svi_dit_set_audio: audio windows must be 16-byte aligned");
    h->aud_first = reinterpret_cast<const bf16*>(audio_first);
    h->aud_latter = reinterpret_cast<const bf16*>(audio_latter);
    h->aud_latter_n = n_latter;
    return SVI_OK;
}

extern "C" svi_status svi_dit_forward(svi_dit* h, const void* x, const float* timestep, const void* context,
                                      const void* clip_feature, const void* y, const void* add_condition, void* out,
                                      int32_t B, int32_t T, int32_t H, int32_t W, int32_t Lc, svi_stream stream) {
    SVI_REQUIRE(h && x && timestep && context && out, "svi_dit_forward: null argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && Lc > 0, "svi_dit_forward: bad sizes");
    SVI_REQUIRE(!h->aud_first || B == 1, "the talk variant takes one sample per forward (audio windows of one clip)");
    SVI_REQUIRE(y || h->cfg.in_dim == 16, "this model takes %d extra input channels: y must be given", h->cfg.in_dim - 16);
    SVI_REQUIRE(!h->cfg.has_image_input || clip_feature, "has_image_input model needs clip_feature");
    const svi_dit_config& c = h->cfg;
    SVI_REQUIRE(T % c.patch_t == 0 && H % c.patch_h == 0 && W % c.patch_w == 0,
                "latent size %dx%dx%d is not divisible by the patch size", T, H, W);
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t thw = (size_t)T * H * W;
    const int L = (T / c.patch_t) * (H / c.patch_h) * (W / c.patch_w);
    for (int b = 0; b < B; ++b) {
        const bf16* xb = reinterpret_cast<const bf16*>(x) + b * 16 * thw;
        const bf16* cb = reinterpret_cast<const bf16*>(context) + (size_t)b * Lc * c.text_dim;
        const bf16* clb = clip_feature ? reinterpret_cast<const bf16*>(clip_feature) + (size_t)b * 257 * 1280 : nullptr;
        const bf16* yb = y ? reinterpret_cast<const bf16*>(y) + b * (size_t)(c.in_dim - 16) * thw : nullptr;
        const bf16* ab = add_condition ? reinterpret_cast<const bf16*>(add_condition) + (size_t)b * L * c.dim : nullptr;
        bf16* ob = reinterpret_cast<bf16*>(out) + b * (size_t)c.out_dim * thw;
        SVI_TRY(forward_one(h, xb, timestep + b, cb, clb, yb, ab, ob, T, H, W, Lc, st));
    }
    return SVI_OK;
}

extern "C" svi_status svi_dit_forward_tea(svi_dit* h, const void* x, const float* timestep, const void* context,
                                          const void* clip_feature, const void* y, const void* add_condition, void* out,
                                          int32_t B, int32_t T, int32_t H, int32_t W, int32_t Lc, int32_t tea_mode, void* residual,
                                          svi_stream stream) {
    SVI_REQUIRE(h && x && timestep && context && out, "svi_dit_forward_tea: null argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && Lc > 0, "svi_dit_forward_tea: bad sizes");
    SVI_REQUIRE(!h->aud_first || B == 1, "the talk variant takes one sample per forward (audio windows of one clip)");
    SVI_REQUIRE(y || h->cfg.in_dim == 16, "this model takes %d extra input channels: y must be given", h->cfg.in_dim - 16);
    SVI_REQUIRE(!h->cfg.has_image_input || clip_feature, "has_image_input model needs clip_feature");
    SVI_REQUIRE(tea_mode >= 0 && tea_mode <= 2 && (tea_mode == 0 || residual), "svi_dit_forward_tea: tea_mode %d needs a residual buffer", tea_mode);
    const svi_dit_config& c = h->cfg;
    SVI_REQUIRE(T % c.patch_t == 0 && H % c.patch_h == 0 && W % c.patch_w == 0,
                "latent size %dx%dx%d is not divisible by the patch size", T, H, W);
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t thw = (size_t)T * H * W;
    const int L = (T / c.patch_t) * (H / c.patch_h) * (W / c.patch_w);
    for (int b = 0; b < B; ++b) {
        const bf16* xb = reinterpret_cast<const bf16*>(x) + b * 16 * thw;
        const bf16* cb = reinterpret_cast<const bf16*>(context) + (size_t)b * Lc * c.text_dim;
        const bf16* clb = clip_feature ? reinterpret_cast<const bf16*>(clip_feature) + (size_t)b * 257 * 1280 : nullptr;
        const bf16* yb = y ? reinterpret_cast<const bf16*>(y) + b * (size_t)(c.in_dim - 16) * thw : nullptr;
        const bf16* ab = add_condition ? reinterpret_cast<const bf16*>(add_condition) + (size_t)b * L * c.dim : nullptr;
        bf16* ob = reinterpret_cast<bf16*>(out) + b * (size_t)c.out_dim * thw;
        bf16* rb = residual ? reinterpret_cast<bf16*>(residual) + (size_t)b * L * c.dim : nullptr;
        SVI_TRY(forward_one(h, xb, timestep + b, cb, clb, yb, ab, ob, T, H, W, Lc, st, tea_mode, rb));
    }
    return SVI_OK;
}

// t_mod = time_projection(silu(time_embedding(sinusoidal(t)))) as bf16 [B, 6, dim] (pipelines/svi_video.py:92-93): the quantity
// TeaCache.check compares between steps (svi_video.py:44-52).
extern "C" svi_status svi_dit_time_mod(svi_dit* h, const float* timestep, void* t_mod_out, int32_t B, svi_stream stream) {
    SVI_REQUIRE(h && timestep && t_mod_out && B > 0, "svi_dit_time_mod: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int D = h->cfg.dim;
    if (!h->ws.base) SVI_TRY(ensure_workspace(h, 256, 8, st));            // the time path only needs the small scratch rows
    for (int b = 0; b < B; ++b) {
        SVI_TRY(stage_time(h, timestep + b, st));
        SVI_CHECK_HIP(hipMemcpyAsync(reinterpret_cast<bf16*>(t_mod_out) + (size_t)b * 6 * D, h->ws.tmod, (size_t)6 * D * 2,
                                     hipMemcpyDeviceToDevice, st));
    }
    return SVI_OK;
}

extern "C" svi_status svi_dit_forward_cfg_pair(svi_dit* h, const void* x, const float* timestep, const void* context_cond,
                                               const void* context_uncond, const void* clip_feature, const void* y,
                                               const void* add_condition, void* out_cond, void* out_uncond, int32_t B, int32_t T,
                                               int32_t H, int32_t W, int32_t Lc, svi_stream stream) {
    SVI_REQUIRE(h && x && timestep && context_cond && context_uncond && out_cond && out_uncond, "svi_dit_forward_cfg_pair: null argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && Lc > 0, "svi_dit_forward_cfg_pair: bad sizes");
    if (h->aud_first) { svi_set_error("svi_dit_forward_cfg_pair: the talk variant's branches differ in their audio; run them as separate forwards"); return SVI_ERR_UNSUPPORTED; }
    SVI_REQUIRE(y || h->cfg.in_dim == 16, "this model takes %d extra input channels: y must be given", h->cfg.in_dim - 16);
    SVI_REQUIRE(!h->cfg.has_image_input || clip_feature, "has_image_input model needs clip_feature");
    const svi_dit_config& c = h->cfg;
    SVI_REQUIRE(T % c.patch_t == 0 && H % c.patch_h == 0 && W % c.patch_w == 0,
                "latent size %dx%dx%d is not divisible by the patch size", T, H, W);
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t thw = (size_t)T * H * W;
    const int L = (T / c.patch_t) * (H / c.patch_h) * (W / c.patch_w);
    for (int b = 0; b < B; ++b) {
        const bf16* xb = reinterpret_cast<const bf16*>(x) + b * 16 * thw;
        const bf16* ca = reinterpret_cast<const bf16*>(context_cond) + (size_t)b * Lc * c.text_dim;
        const bf16* cbn = reinterpret_cast<const bf16*>(context_uncond) + (size_t)b * Lc * c.text_dim;
        const bf16* clb = clip_feature ? reinterpret_cast<const bf16*>(clip_feature) + (size_t)b * 257 * 1280 : nullptr;
        const bf16* yb = y ? reinterpret_cast<const bf16*>(y) + b * (size_t)(c.in_dim - 16) * thw : nullptr;
        const bf16* ab = add_condition ? reinterpret_cast<const bf16*>(add_condition) + (size_t)b * L * c.dim : nullptr;
        bf16* oa = reinterpret_cast<bf16*>(out_cond) + b * (size_t)c.out_dim * thw;
        bf16* ob = reinterpret_cast<bf16*>(out_uncond) + b * (size_t)c.out_dim * thw;
        SVI_TRY(forward_pair(h, xb, timestep + b, ca, cbn, clb, yb, ab, oa, ob, T, H, W, Lc, st));
    }
    return SVI_OK;
}

// A rolling window hands the NEXT clip's prompt embedding (and CLIP feature) to the model in the SAME device tensors (the caller copied the new
// values into them): the entry keyed by these pointers is recomputed IN PLACE — projected context, identical-suffix summary, every block's
// cross-attention K / V^T — in the buffers it already owns.  No address moves and svi_dit_generation stays where it is, so a hipGraph of the step
// captured for the previous clip (which reads exactly these buffers) replays on the new prompt.  Stream-ordered: enqueue it on the stream the
// replays run on.  Without an entry for the key it is the first fill (allocation; the generation moves, as in a forward's miss).
extern "C" svi_status svi_dit_context_refill(svi_dit* h, const void* context, const void* clip_feature, int32_t Lc, svi_stream stream) {
    SVI_REQUIRE(h && context, "svi_dit_context_refill: null argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(h->ctx_cache_on, "svi_dit_context_refill: the context cache is off (svi_dit_context_cache(h, 1) first)");
    SVI_REQUIRE(Lc > 0, "svi_dit_context_refill: bad context length %d", Lc);
    const svi_dit_config& c = h->cfg;
    SVI_REQUIRE(!c.has_image_input || clip_feature, "has_image_input model needs clip_feature");
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bf16* ctx = reinterpret_cast<const bf16*>(context);
    const bf16* clip = c.has_image_input ? reinterpret_cast<const bf16*>(clip_feature) : nullptr;
    Workspace& w = h->ws;
    if (!w.base || w.Lc != Lc) SVI_TRY(ensure_workspace(h, w.L > 0 ? w.L : 8, Lc, st));      // the projection's scratch rows (CTXH, IMG*) are laid out per Lc
    CtxUse cu{};
    SVI_TRY(stage_context(h, ctx, clip, /*y: only its presence is checked*/ ctx, Lc, &cu, st));
    SVI_REQUIRE(cu.ce != nullptr, "svi_dit_context_refill: no cache entry");
    if (!cu.compute) {                                                                           // a hit: the same buffers, new contents
        const int before = cu.ce->key_blocks;
        SVI_TRY(project_context(h, ctx, clip, Lc, cu.CTXp, cu.ce->tail, st));
        cu.ce->key_blocks = cu.key_blocks = (svi_switches().cross_fused && svi_switches().cross_dedup) ? read_key_blocks(cu.ce->tail, st) : 0;
        if (cu.ce->key_blocks != before) ++h->generation;     // a captured step holds cross-attention launches made for the OLD key-block count (svi_launch_flash_cross): it must not be replayed
    }
    for (int l = 0; l < c.num_layers; ++l) {
        CtxKV kv = kv_of(h, cu, l);
        SVI_TRY(fill_block_kv(h, l, cu.CTXp, Lc, kv, st));
    }
    return SVI_OK;
}

// ------------------------------------------------------------------------------------------------
// Sequence-parallel (Ulysses) execution of one forward: SURVEY §8e axis 3, the reference's USP path
// (pipelines/svi_video.py:119-135 chunk / all_gather, distributed/xdit_context_parallel.py usp_attn_forward).  A rank owns the
// token rows [row0, row0 + nrows) of the (f h w) sequence for everything that is row-local (norms, projections, cross
// attention, MLP, head) and trades tokens for heads around self-attention.  The exchange itself (RCCL all-to-all) is the
// caller's (svi_hip/sequence_parallel.py); these entry points are the row-local pieces between the exchanges.
// ------------------------------------------------------------------------------------------------
extern "C" svi_status svi_dit_sp_begin(svi_dit* h, const void* x, const float* timestep, const void* context,
                                       const void* clip_feature, const void* y, const void* add_condition, int32_t T, int32_t H,
                                       int32_t W, int32_t Lc, int32_t row0, int32_t nrows, svi_stream stream) {
    SVI_REQUIRE(h && x && timestep && context, "svi_dit_sp_begin: null argument");
    SVI_REQUIRE_DEVICE(h);
    const svi_dit_config& c = h->cfg;
    SVI_REQUIRE(T > 0 && H > 0 && W > 0 && Lc > 0 && T % c.patch_t == 0 && H % c.patch_h == 0 && W % c.patch_w == 0, "svi_dit_sp_begin: bad sizes");
    if (h->aud_first) { svi_set_error("svi_dit_sp_begin: the talk variant's per-frame audio attention is not served on sequence shards"); return SVI_ERR_UNSUPPORTED; }
    SVI_REQUIRE(y || c.in_dim == 16, "this model takes %d extra input channels: y must be given", c.in_dim - 16);
    SVI_REQUIRE(!c.has_image_input || clip_feature, "has_image_input model needs clip_feature");
    const int f = T / c.patch_t, hh = H / c.patch_h, ww = W / c.patch_w, L = f * hh * ww;
    SVI_REQUIRE(row0 >= 0 && nrows > 0 && row0 + nrows <= L, "svi_dit_sp_begin: rows [%d, %d) outside the %d-token sequence", row0, row0 + nrows, L);
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SVI_TRY(ensure_workspace(h, nrows, Lc, st));
    SVI_TRY(ensure_rope(h, f, hh, ww));
    SVI_TRY(stage_time(h, timestep, st));
    CtxUse cu{};
    SVI_TRY(stage_context(h, reinterpret_cast<const bf16*>(context), reinterpret_cast<const bf16*>(clip_feature),
                          reinterpret_cast<const bf16*>(y), Lc, &cu, st));
    SVI_TRY(stage_embed(h, reinterpret_cast<const bf16*>(x), reinterpret_cast<const bf16*>(y), reinterpret_cast<const bf16*>(add_condition),
                        T, H, W, nrows, st, row0));
    h->sp_rows = nrows; h->sp_row0 = row0; h->sp_Lc = Lc; h->sp_active = true; h->sp_ctxp = cu.CTXp; h->sp_nb = 1;
    h->sp_kv.clear();
    for (int l = 0; l < c.num_layers; ++l) h->sp_kv.push_back(kv_of(h, cu, l));
    return SVI_OK;
}

// Both forwards of a CFG step on ONE sequence shard, stacked (the single-rank forward_pair's form on a rank's rows): X holds the conditional branch's
// nrows rows on top of the unconditional branch's; every row-local kernel of a block (norms, projections, MLP, head) then runs once over 2 nrows rows —
// a quarter-size shard's launches are half-size again — and the exchange moves each branch's q | k / output pieces as their own contiguous blocks
// (SviScatter::rows_per_sample).  No CFG exchange between ranks: a rank ends with both branches' head rows.  Needs the context cache (each prompt's
// projected context and K / V in buffers of its own).  The results are bit-identical to two svi_dit_sp_begin forwards.
extern "C" svi_status svi_dit_sp_begin_pair(svi_dit* h, const void* x, const float* timestep, const void* context_cond, const void* context_uncond,
                                            const void* clip_feature, const void* y, const void* add_condition, int32_t T, int32_t H,
                                            int32_t W, int32_t Lc, int32_t row0, int32_t nrows, svi_stream stream) {
    SVI_REQUIRE(h && x && timestep && context_cond && context_uncond && context_cond != context_uncond, "svi_dit_sp_begin_pair: null argument (or one prompt given twice)");
    SVI_REQUIRE_DEVICE(h);
    const svi_dit_config& c = h->cfg;
    SVI_REQUIRE(h->ctx_cache_on, "svi_dit_sp_begin_pair: the stacked CFG pair needs the context cache (svi_dit_context_cache(h, 1))");
    SVI_REQUIRE(T > 0 && H > 0 && W > 0 && Lc > 0 && T % c.patch_t == 0 && H % c.patch_h == 0 && W % c.patch_w == 0, "svi_dit_sp_begin_pair: bad sizes");
    if (h->aud_first) { svi_set_error("svi_dit_sp_begin_pair: the talk variant's per-frame audio attention is not served on sequence shards"); return SVI_ERR_UNSUPPORTED; }
    SVI_REQUIRE(y || c.in_dim == 16, "this model takes %d extra input channels: y must be given", c.in_dim - 16);
    SVI_REQUIRE(!c.has_image_input || clip_feature, "has_image_input model needs clip_feature");
    const int f = T / c.patch_t, hh = H / c.patch_h, ww = W / c.patch_w, L = f * hh * ww;
    SVI_REQUIRE(row0 >= 0 && nrows > 0 && row0 + nrows <= L, "svi_dit_sp_begin_pair: rows [%d, %d) outside the %d-token sequence", row0, row0 + nrows, L);
    const size_t widest = (size_t)std::max(c.ffn_dim, 2 * c.dim);
    SVI_REQUIRE((size_t)2 * nrows * widest * 2 < ((size_t)1 << 31), "svi_dit_sp_begin_pair: 2 x %d rows of the widest activation reach 2 GiB: run the branches separately", nrows);
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SVI_TRY(ensure_workspace(h, 2 * nrows, Lc, st));
    SVI_TRY(ensure_rope(h, f, hh, ww));
    SVI_TRY(stage_time(h, timestep, st));
    Workspace& w = h->ws;
    SVI_TRY(stage_embed(h, reinterpret_cast<const bf16*>(x), reinterpret_cast<const bf16*>(y), reinterpret_cast<const bf16*>(add_condition),
                        T, H, W, nrows, st, row0));
    SVI_CHECK_HIP(hipMemcpyAsync(w.X + (size_t)nrows * c.dim, w.X, (size_t)nrows * c.dim * 2, hipMemcpyDeviceToDevice, st));
    CtxUse cu[2]{};
    const bf16* ctxs[2] = {reinterpret_cast<const bf16*>(context_cond), reinterpret_cast<const bf16*>(context_uncond)};
    for (int k = 0; k < 2; ++k) {
        SVI_TRY(stage_context(h, ctxs[k], reinterpret_cast<const bf16*>(clip_feature), reinterpret_cast<const bf16*>(y), Lc, &cu[k], st));
        SVI_REQUIRE(cu[k].ce != nullptr, "svi_dit_sp_begin_pair: no cache entry");
    }
    h->sp_rows = nrows; h->sp_row0 = row0; h->sp_Lc = Lc; h->sp_active = true; h->sp_nb = 2;
    h->sp_ctxp = cu[0].CTXp; h->sp_ctxp_b = cu[1].CTXp;
    h->sp_kv.clear(); h->sp_kv_b.clear();
    for (int l = 0; l < c.num_layers; ++l) { h->sp_kv.push_back(kv_of(h, cu[0], l)); h->sp_kv_b.push_back(kv_of(h, cu[1], l)); }
    return SVI_OK;
}

extern "C" svi_status svi_dit_sp_block_qkv(svi_dit* h, int32_t layer, void* q_send, void* k_send, void* vt_out, int32_t ldvt, int32_t P, int32_t G,
                                           svi_stream stream) {
    return svi_dit_sp_block_qkv_part(h, layer, q_send, k_send, vt_out, ldvt, P, G, 0, stream);
}
extern "C" svi_status svi_dit_sp_block_qkv_part(svi_dit* h, int32_t layer, void* q_send, void* k_send, void* vt_out, int32_t ldvt, int32_t P, int32_t G,
                                                int32_t part, svi_stream stream) {
    SVI_REQUIRE(h && h->sp_active && q_send && k_send && vt_out && part >= 0 && part <= 2, "svi_dit_sp_block_qkv: no shard in flight (svi_dit_sp_begin), null buffer or bad part");
    SVI_REQUIRE_DEVICE(h);
    const int D = h->cfg.dim;
    const int nb = h->sp_nb, rows = nb * h->sp_rows;
    SVI_REQUIRE(layer >= 0 && layer < h->cfg.num_layers && ldvt >= rows && ldvt % 8 == 0, "svi_dit_sp_block_qkv: bad layer / ldvt");
    SVI_REQUIRE(P > 0 && G > 0 && h->cfg.num_heads % (P * G) == 0, "svi_dit_sp_block_qkv: %d heads do not split into %d ranks x %d head groups", h->cfg.num_heads, P, G);
    // stacked pair: a token's two branches side by side in the send block, [G][P][nrows][2][Dg] (SviScatter)
    SviScatter sc{reinterpret_cast<bf16*>(q_send), reinterpret_cast<bf16*>(k_send), P, D / P, D / P / G, nb > 1 ? h->sp_rows : 0};
    return block_qkv(h, layer, h->ws.X, h->ws.modf + (size_t)layer * 6 * D, rows, h->sp_row0, h->ws.QK, reinterpret_cast<bf16*>(vt_out), ldvt,
                     reinterpret_cast<hipStream_t>(stream), &sc, nb, part);
}

extern "C" svi_status svi_sp_unpack_vt(const void* recv, void* out, int32_t P, int32_t Dp, int32_t Ls, int32_t lds, int32_t L8, int32_t nb, int32_t Dg, svi_stream stream) {
    SVI_REQUIRE(recv && out, "svi_sp_unpack_vt: null argument");
    return svi_launch_sp_unpack_vt(reinterpret_cast<const bf16*>(recv), reinterpret_cast<bf16*>(out), P, Dp, Ls, lds, L8, reinterpret_cast<hipStream_t>(stream), nb, Dg);
}
extern "C" svi_status svi_sp_unpack_out(const void* recv, void* out, int32_t P, int32_t G, int32_t Ls, int32_t Dg, int32_t nb, svi_stream stream) {
    SVI_REQUIRE(recv && out, "svi_sp_unpack_out: null argument");
    return svi_launch_sp_unpack_out(reinterpret_cast<const bf16*>(recv), reinterpret_cast<bf16*>(out), P, G, Ls, Dg, reinterpret_cast<hipStream_t>(stream), nb);
}

// TeaCache inside a sequence-parallel forward (the reference allows the combination: svi_video.py:112-131 checks on the full x, then
// chunks it, and store / update act on the rank's chunk).  mode 0: snapshot the shard's rows before the blocks; 1: residual =
// bf16(x_after - x_before) of the shard's rows (TeaCache.store, :64-66); 2: x = bf16(x + residual) in place of the blocks (:68-70).
extern "C" svi_status svi_dit_sp_tea(svi_dit* h, int32_t mode, void* residual, svi_stream stream) {
    SVI_REQUIRE(h && h->sp_active && mode >= 0 && mode <= 2 && (mode == 0 || residual), "svi_dit_sp_tea: no shard in flight, bad mode or null residual");
    SVI_REQUIRE_DEVICE(h);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Workspace& w = h->ws;
    const int64_t n = (int64_t)h->sp_nb * h->sp_rows * h->cfg.dim;
    if (mode == 0) { SVI_CHECK_HIP(hipMemcpyAsync(w.X2, w.X, (size_t)n * 2, hipMemcpyDeviceToDevice, st)); return SVI_OK; }
    if (mode == 1) return svi_launch_sub_bf16(reinterpret_cast<bf16*>(residual), w.X, w.X2, n, st);
    return svi_launch_add_bf16(w.X, reinterpret_cast<const bf16*>(residual), n, st);
}

extern "C" svi_status svi_dit_sp_block_rest(svi_dit* h, int32_t layer, const void* attn, svi_stream stream) {
    SVI_REQUIRE(h && h->sp_active && attn, "svi_dit_sp_block_rest: no shard in flight or null buffer");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(layer >= 0 && layer < h->cfg.num_layers, "svi_dit_sp_block_rest: bad layer");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const float* modf = h->ws.modf + (size_t)layer * 6 * h->cfg.dim;
    if (h->sp_nb == 2) {
        SVI_TRY(block_attn_out(h, layer, h->ws.X, reinterpret_cast<const bf16*>(attn), modf, 2 * h->sp_rows, st, 2));
        const bf16* CTXs[2] = {h->sp_ctxp, h->sp_ctxp_b};
        CtxKV kvs[2] = {h->sp_kv[layer], h->sp_kv_b[layer]};
        return run_block_rest_n(h, layer, h->ws.X, CTXs, modf, h->sp_rows, h->sp_Lc, kvs, 2, st);
    }
    SVI_TRY(block_attn_out(h, layer, h->ws.X, reinterpret_cast<const bf16*>(attn), modf, h->sp_rows, st));
    return run_block_rest(h, layer, h->ws.X, h->sp_ctxp, modf, h->sp_rows, h->sp_Lc, h->sp_kv[layer], st);
}

extern "C" svi_status svi_dit_sp_head(svi_dit* h, void* head_rows_out, svi_stream stream) {
    SVI_REQUIRE(h && h->sp_active && head_rows_out, "svi_dit_sp_head: no shard in flight or null buffer");
    SVI_REQUIRE_DEVICE(h);
    h->sp_active = false;
    return stage_head_rows(h, reinterpret_cast<bf16*>(head_rows_out), h->sp_nb * h->sp_rows, reinterpret_cast<hipStream_t>(stream), h->sp_nb);
}

extern "C" svi_status svi_dit_unpatchify(svi_dit* h, const void* head_rows, void* out, int32_t T, int32_t H, int32_t W, svi_stream stream) {
    SVI_REQUIRE(h && head_rows && out && T > 0 && H > 0 && W > 0, "svi_dit_unpatchify: bad argument");
    SVI_REQUIRE_DEVICE(h);
    return stage_unpatchify(h, reinterpret_cast<const bf16*>(head_rows), reinterpret_cast<bf16*>(out), T, H, W, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int32_t svi_dit_head_ld(svi_dit* h) { return h ? head_ld(h->cfg) : 0; }

// A counter that moves whenever device state a captured hipGraph of this handle's forwards may have baked in stops being valid:
// the workspace was (re)allocated, a context-cache entry was filled / evicted, the cache was reset, a weight was re-bound.
// A replay is only legal while the value equals the one read right after the capture.
extern "C" int64_t svi_dit_generation(svi_dit* h) { return h ? (int64_t)(h->generation + svi_stream_buffer_generation()) : -1; }

extern "C" svi_status svi_dit_block_forward(svi_dit* h, int32_t layer, void* x_inout, const void* context,
                                            const void* t_mod, int32_t f, int32_t hh, int32_t ww, int32_t Lc,
                                            svi_stream stream) {
    SVI_REQUIRE(h && x_inout && context && t_mod, "svi_dit_block_forward: null argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(layer >= 0 && layer < h->cfg.num_layers, "layer %d out of range", layer);
    SVI_REQUIRE(f > 0 && hh > 0 && ww > 0 && Lc > 0, "bad grid");
    SVI_TRY(svi_dit_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int L = f * hh * ww, D = h->cfg.dim;
    SVI_TRY(ensure_workspace(h, L, Lc, st));
    SVI_TRY(ensure_rope(h, f, hh, ww));
    float* modf = h->ws.modf + (size_t)layer * 6 * D;
    SVI_TRY(mod_one(h->blocks[layer].modulation, reinterpret_cast<const bf16*>(t_mod), modf, D, 6, (1 << 1) | (1 << 4), 6, st));
    CtxKV kv{h->ws.CK, h->ws.CVT, h->ws.CKi, h->ws.CVTi, true, nullptr, 0};      // the context arrives projected: no statement about its input rows
    return run_block(h, layer, reinterpret_cast<bf16*>(x_inout), reinterpret_cast<const bf16*>(context), modf, L, Lc, kv, st);
}
