// svi_encoders.hip — the two prompt-side encoders of the Wan pipelines: they run once per clip, outside the step
// loop, and produce the `context` / `clip_feature` operands of the DiT.
//
//   svi_t5    WanTextEncoder (models/wan_video_text_encoder.py:209-256): umT5 encoder, bf16 as the pipeline keeps it
//             (24 x [T5LayerNorm, attention with a per-layer relative position bias and no 1/sqrt(d) scale, gated-GELU FFN]).
//   svi_clip  WanImageEncoder.encode_image (models/wan_video_image_encoder.py:864-880): bicubic resize + CLIP normalisation +
//             the first L-1 blocks of the ViT-H/14 visual tower, fp32 as SVI runs it (pipelines/svi_video.py:307-309).
//
// All projections go through the MFMA GEMMs of the hot path (svi_launch_gemm bf16, svi_launch_gemm_f32 exact fp32); what is
// written here are the row kernels around them and one small attention kernel.  The sequences are at most 512 keys x 64 channels
// per head and the whole attention of either encoder is < 2 % of its FLOP, so that kernel is a plain LDS-staged VALU kernel whose
// point is to restate the reference's rounding points exactly (scores rounded to bf16, bias added in bf16, softmax in fp32 over the
// whole row, NORMALISED probabilities rounded to bf16 before P.V) — which an online-softmax kernel cannot do.
//
// Design choice (T5): padded positions are masked out as KEYS by the reference and their own output rows are zeroed by the caller
// (prompters/wan_prompter.py:108-112), so only the first `rows` (>= number of valid tokens) query rows are computed; rows = L gives
// the reference's text_encoder(ids, mask) output in full.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "svi_common.h"

namespace {

// ============================================================ shared row kernels ==================================================
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16>(const bf16* p) { return (float)*p; }

// One workgroup = one head x ENC_RQ query rows.  LDS: the query rows (fp32) and the full score rows (fp32).
//   BF16PTS: restate the bf16 module's rounding points (t5:73-76); otherwise plain fp32 (CLIP, SDPA in fp32).
//   bias:  emb[bucket_tab[key - query + tab_zero]][head]  (bf16 table [buckets][heads]) or none.
#define ENC_RQ 16
template <typename T, int D, bool BF16PTS>
__global__ __launch_bounds__(256) void enc_attention_kernel(const T* __restrict__ Q, int ldq, const T* __restrict__ K, int ldk,
                                                            const T* __restrict__ V, int ldv, T* __restrict__ O, int ldo, int Lq, int Lk,
                                                            float scale, const bf16* __restrict__ emb, int heads,
                                                            const int* __restrict__ bucket_tab, int tab_zero) {
    extern __shared__ __attribute__((aligned(16))) float enc_smem[];
    float* Qs = enc_smem;                       // [ENC_RQ][D]
    float* S = enc_smem + ENC_RQ * D;           // [ENC_RQ][Lk]
    const int tid = threadIdx.x, head = blockIdx.y, q0 = blockIdx.x * ENC_RQ;
    const int nq = min(ENC_RQ, Lq - q0);
    for (int i = tid; i < ENC_RQ * D; i += 256) {
        const int r = i / D, c = i - r * D;
        Qs[i] = r < nq ? ldf(Q + (size_t)(q0 + r) * ldq + head * D + c) : 0.f;
    }
    __syncthreads();
    // ---- scores: thread <-> key
    for (int j = tid; j < Lk; j += 256) {
        float acc[ENC_RQ];
#pragma unroll
        for (int r = 0; r < ENC_RQ; ++r) acc[r] = 0.f;
        const T* kp = K + (size_t)j * ldk + head * D;
#pragma unroll 4
        for (int c = 0; c < D; c += 4) {
            float kv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) kv[e] = ldf(kp + c + e);
#pragma unroll
            for (int r = 0; r < ENC_RQ; ++r) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(Qs + r * D + c);
                acc[r] = fmaf(qv[0], kv[0], acc[r]); acc[r] = fmaf(qv[1], kv[1], acc[r]);
                acc[r] = fmaf(qv[2], kv[2], acc[r]); acc[r] = fmaf(qv[3], kv[3], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < ENC_RQ; ++r) {
            float s = acc[r] * scale;
            if (BF16PTS) s = rbf(s);
            if (emb && r < nq) {                  // rows past the last query have no table entry (and no output)
                const float b = (float)emb[(size_t)bucket_tab[j - (q0 + r) + tab_zero] * heads + head];
                s += b;
                if (BF16PTS) s = rbf(s);
            }
            S[r * Lk + j] = s;
        }
    }
    __syncthreads();
    // ---- softmax over the whole row, fp32 (t5:75); wave w owns rows 4w..4w+3
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int r = 4 * wave; r < 4 * wave + 4; ++r) {
            float* row = S + r * Lk;
            float m = -INFINITY;
            for (int j = lane; j < Lk; j += 64) m = fmaxf(m, row[j]);
            m = wave_max(m);
            float sum = 0.f;
            for (int j = lane; j < Lk; j += 64) { const float e = expf(row[j] - m); row[j] = e; sum += e; }
            sum = wave_sum(sum);
            for (int j = lane; j < Lk; j += 64) { const float pr = row[j] / sum; row[j] = BF16PTS ? rbf(pr) : pr; }
        }
    }
    __syncthreads();
    // ---- O = P V: thread <-> (row group, channel); a V element is loaded once and used for all rows of the group
    constexpr int NG = 256 / D, RPG = (ENC_RQ + NG - 1) / NG;
    const int g = tid / D, c = tid - g * D;
    if (g < NG) {
        float acc[RPG];
#pragma unroll
        for (int r = 0; r < RPG; ++r) acc[r] = 0.f;
        const T* vp = V + head * D + c;
        for (int j = 0; j < Lk; ++j) {
            const float v = ldf(vp + (size_t)j * ldv);
#pragma unroll
            for (int r = 0; r < RPG; ++r) {
                const int row = g * RPG + r;
                if (row < ENC_RQ) acc[r] = fmaf(S[row * Lk + j], v, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
            const int row = g * RPG + r;
            if (row < nq) O[(size_t)(q0 + row) * ldo + head * D + c] = (T)acc[r];
        }
    }
}

template <typename T, bool BF16PTS>
svi_status launch_enc_attention(const T* Q, int ldq, const T* K, int ldk, const T* V, int ldv, T* O, int ldo, int Lq, int Lk, int heads, int D,
                                float scale, const bf16* emb, const int* bucket_tab, int tab_zero, hipStream_t st) {
    SVI_REQUIRE(Lq > 0 && Lk > 0 && Lk <= 2048, "encoder attention: %d keys (supported: 1..2048)", Lk);
    const int lds = (ENC_RQ * D + ENC_RQ * Lk) * 4;
    dim3 grid((Lq + ENC_RQ - 1) / ENC_RQ, heads), block(256);
#define ENC_CASE(DD)                                                                                                               \
    case DD:                                                                                                                       \
        SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(enc_attention_kernel<T, DD, BF16PTS>), 160 * 1024 - 256));              \
        hipLaunchKernelGGL((enc_attention_kernel<T, DD, BF16PTS>), grid, block, lds, st, Q, ldq, K, ldk, V, ldv, O, ldo, Lq, Lk, scale, \
                           emb, heads, bucket_tab, tab_zero);                                                                      \
        break;
    switch (D) {
        ENC_CASE(32) ENC_CASE(64) ENC_CASE(80) ENC_CASE(128)
        default: svi_set_error("encoder attention: head dim %d (supported: 32, 64, 80, 128)", D); return SVI_ERR_UNSUPPORTED;
    }
#undef ENC_CASE
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ============================================================ T5 kernels ==========================================================
// token_embedding (t5:240): rows of the table
__global__ void t5_embed_kernel(const int64_t* __restrict__ ids, const bf16* __restrict__ table, bf16* __restrict__ x, int rows, int dim, int vocab) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // the host wrapper rejects out-of-range ids; never read outside the table
    const bf16x8* src = reinterpret_cast<const bf16x8*>(table + (size_t)id * dim);
    bf16x8* dst = reinterpret_cast<bf16x8*>(x + (size_t)r * dim);
    for (int i = threadIdx.x; i < dim / 8; i += blockDim.x) dst[i] = src[i];
}

// T5LayerNorm in a bf16 module (t5:31-35): bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) ), statistics in fp32.  One wave per row.
__global__ __launch_bounds__(256) void t5_norm_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, const bf16* __restrict__ w, int rows,
                                                      int dim, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bf16* xp = x + (size_t)row * dim;
    float ss = 0.f;
    for (int i = lane * 8; i < dim; i += 512) {
        const bf16x8 v = ld_bf16x8(xp + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)dim + eps);
    bf16* op = out + (size_t)row * dim;
    for (int i = lane * 8; i < dim; i += 512) {
        const bf16x8 v = ld_bf16x8(xp + i), ww = ld_bf16x8(w + i);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)ww[e] * rbf((float)v[e] * r));
        st_bf16x8(op + i, o);
    }
}

// fc1(x) * GELU(gate(x)) with the reference's own GELU (t5:16-20), evaluated op by op in bf16 as a bf16 module does:
//   0.5 * x * (1.0 + tanh( sqrt(2/pi) * (x + 0.044715 * pow(x, 3)) ))
__global__ void t5_gated_gelu_kernel(const bf16* __restrict__ gate, const bf16* __restrict__ fc1, bf16* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    const bf16x8 g = ld_bf16x8(gate + i), f = ld_bf16x8(fc1 + i);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (float)g[e];
        const float a = rbf(0.5f * x);
        const float p = rbf(x * x * x);                     // exact in fp32 (3 x 8 significant bits), then the bf16 rounding of torch.pow
        const float b = rbf(0.044715f * p);
        const float c = rbf(x + b);
        const float d = rbf(0.7978845608028654f * c);
        const float t = rbf(tanhf(d));
        const float u = rbf(1.0f + t);
        const float ge = rbf(a * u);
        o[e] = (bf16)((float)f[e] * ge);
    }
    st_bf16x8(out + i, o);
}

__global__ void zero_bf16_kernel(bf16* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (bf16)0.f;
}

// ============================================================ CLIP kernels ========================================================
// cubic convolution coefficients, A = -0.75 (torch upsample_bicubic2d)
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = x2 + 1.0f;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
// encode_image preprocessing (image_encoder:866-873) fused with the patch gather of the patch-embedding convolution:
//   bicubic (align_corners=False, no antialias) [3,H,W] -> [3,S,S];  v*0.5+0.5;  (v - mean) / std;  then patch (py,px), element (c,ky,kx)
//   goes to patches[py*np + px][c*ps*ps + ky*ps + kx]  (the Conv2d weight [dim,3,ps,ps] flattened is the matching [dim, 3*ps*ps] matrix)
#pragma clang fp contract(off)
__global__ void clip_preprocess_kernel(const float* __restrict__ img, int H, int W, float* __restrict__ patches, int S, int ps, int ldp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * S * S) return;
    const int c = i / (S * S), oy = (i / S) % S, ox = i % S;
    const float sy = (float)H / (float)S, sx = (float)W / (float)S;
    const float ry = sy * ((float)oy + 0.5f) - 0.5f, rx = sx * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    float wy[4], wx[4];
    cubic_coeffs(ry - fy, wy);
    cubic_coeffs(rx - fx, wx);
    const float* src = img + (size_t)c * H * W;
    float v = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), H - 1);
        float rowv = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), W - 1);
            rowv += src[(size_t)yy * W + xx] * wx[b];
        }
        v += rowv * wy[a];
    }
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, sd[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    v = v * 0.5f + 0.5f;
    v = (v - mean[c]) / sd[c];
    const int np = S / ps, py = oy / ps, ky = oy - py * ps, px = ox / ps, kx = ox - px * ps;
    patches[(size_t)(py * np + px) * ldp + (c * ps + ky) * ps + kx] = v;
}
#pragma clang fp contract(fast)

// row 0 of the token matrix: cls_embedding + pos_embedding[0]  (image_encoder:460-466)
__global__ void clip_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dim) x[i] = cls[i] + pos[i];
}

// nn.LayerNorm in fp32 (biased variance), one wave per row
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ w,
                                                            const float* __restrict__ b, int rows, int dim, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xp = x + (size_t)row * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += xp[i];
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
    for (int i = lane; i < dim; i += 64) { const float d = xp[i] - mean; q = fmaf(d, d, q); }
    const float r = rsqrtf(wave_sum(q) / (float)dim + eps);
    float* op = out + (size_t)row * dim;
    for (int i = lane; i < dim; i += 64) op[i] = (xp[i] - mean) * r * w[i] + b[i];
}

__global__ void gelu_erf_f32_kernel(float* __restrict__ x, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = gelu_erf_f(x[i]);
}

struct Param { const void* p = nullptr; std::vector<int64_t> shape; };

bool shape_is(const Param& q, std::initializer_list<int64_t> want) {
    if (q.shape.size() != want.size()) return false;
    size_t i = 0;
    for (int64_t w : want) if (q.shape[i++] != w) return false;
    return true;
}

svi_status grow(char** buf, size_t* have, size_t need, const char* what) {
    if (*have >= need) return SVI_OK;
    if (*buf) { SVI_CHECK_HIP(hipFree(*buf)); *buf = nullptr; *have = 0; }
    hipError_t e = hipMalloc((void**)buf, need);
    if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B %s) failed: %s", need, what, hipGetErrorString(e)); return SVI_ERR_OOM; }
    *have = need;
    return SVI_OK;
}

}  // namespace

// ================================================================= T5 ==============================================================
struct svi_t5 {
    int device = -1;
    svi_t5_config cfg{};
    std::map<std::string, Param> w;
    char* ws = nullptr; size_t ws_bytes = 0;
    int* tab = nullptr; int tab_len = 0;             // relative-position buckets for offsets -(tab_len-1) .. tab_len-1
    int tab_host = 0;                                // built in the host's arithmetic (SVI_T5_BUCKETS=host) or the device's
};

// T5RelativeEmbedding._relative_position_bucket (t5:175-194), bidirectional, for rel = key - query, on the host in the fp32 steps of the
// reference's tensor ops.
extern "C" svi_status svi_t5_relative_buckets(int32_t num_buckets, int32_t max_dist, int32_t len, int32_t* out) {
    SVI_REQUIRE(out && num_buckets >= 4 && num_buckets % 2 == 0 && max_dist > num_buckets / 4 && len > 0, "svi_t5_relative_buckets: bad argument");
    const int nb = num_buckets / 2, max_exact = nb / 2;
    const float denom = (float)log((double)max_dist / (double)max_exact);          // math.log(...) of Python floats, then an fp32 tensor op
    for (int rel = -(len - 1); rel <= len - 1; ++rel) {
        int b = rel > 0 ? nb : 0;
        const int a = abs(rel);
        if (a < max_exact) b += a;
        else {
            const float v = logf((float)a / (float)max_exact) / denom * (float)(nb - max_exact);
            int large = max_exact + (int)v;
            if (large > nb - 1) large = nb - 1;
            b += large;
        }
        out[rel + len - 1] = b;
    }
    return SVI_OK;
}

// The same table built ON THE DEVICE, in the arithmetic the reference's tensor ops perform there: the module evaluates
// _relative_position_bucket on the embedding's device (t5:160-165), and a CUDA/HIP tensor divided by a Python scalar is multiplied
// by the scalar's fp32 reciprocal (ATen BinaryDivTrueKernel), the logarithm is the device's logf.  At offsets where the ratio is an
// exact integer (|rel| = 16, 32, 64 for the Wan configuration) the host's division and the device's reciprocal-multiply can fall on
// different sides of it, which selects a different bias row — so the table a GPU forward uses is made by the GPU's arithmetic.
__global__ void t5_bucket_kernel(int nb, int max_exact, float inv_exact, float inv_denom, float span, int len, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * len - 1) return;
    const int rel = i - (len - 1);
    int b = rel > 0 ? nb : 0;
    const int a = rel < 0 ? -rel : rel;
    if (a < max_exact) b += a;
    else {
        const float x = (float)a * inv_exact;              // rel_pos.float() / max_exact
        float v = logf(x) * inv_denom;                     // torch.log(.) / math.log(max_dist / max_exact)
        v = v * span;                                      // * (num_buckets - max_exact)
        long large = (long)max_exact + (long)v;            // .long() truncates toward zero
        if (large > nb - 1) large = nb - 1;
        b += (int)large;
    }
    out[i] = b;
}
static svi_status t5_device_table(svi_t5* h, int L) {
    const int want_host = svi_switches().t5_host_buckets;
    if (h->tab_len >= L && h->tab_host == want_host) return SVI_OK;
    const svi_t5_config& c = h->cfg;
    if (h->tab) { SVI_CHECK_HIP(hipFree(h->tab)); h->tab = nullptr; h->tab_len = 0; }
    SVI_CHECK_HIP(hipMalloc((void**)&h->tab, (size_t)(2 * L - 1) * 4));
    h->tab_host = want_host;
    if (want_host) {           // SVI_T5_BUCKETS=host: the CPU module's arithmetic (tests against CPU-made fixtures)
        std::vector<int32_t> host((size_t)2 * L - 1);
        SVI_TRY(svi_t5_relative_buckets(c.num_buckets, c.max_dist, L, host.data()));
        SVI_CHECK_HIP(hipMemcpy(h->tab, host.data(), host.size() * 4, hipMemcpyHostToDevice));
        SVI_CHECK_HIP(hipDeviceSynchronize());
        h->tab_len = L;
        return SVI_OK;
    }
    const int nb = c.num_buckets / 2, max_exact = nb / 2;
    const float inv_exact = 1.0f / (float)max_exact;
    const float inv_denom = 1.0f / (float)log((double)c.max_dist / (double)max_exact);
    hipLaunchKernelGGL(t5_bucket_kernel, dim3((2 * L - 1 + 255) / 256), dim3(256), 0, nullptr, nb, max_exact, inv_exact, inv_denom,
                       (float)(nb - max_exact), L, h->tab);
    SVI_LAUNCH_CHECK();
    SVI_CHECK_HIP(hipDeviceSynchronize());                  // allocation-time work on the null stream: the caller's stream need not be ordered after it
    h->tab_len = L;
    return SVI_OK;
}
// The device-built table for offsets -(len-1) .. len-1 (tests compare it with the reference formula evaluated by torch on the device).
extern "C" svi_status svi_t5_device_buckets(svi_t5* h, int32_t len, int32_t* out) {
    SVI_REQUIRE(h && out && len > 0 && len <= 2048, "svi_t5_device_buckets: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_TRY(t5_device_table(h, len));
    std::vector<int32_t> host((size_t)2 * h->tab_len - 1);
    SVI_CHECK_HIP(hipMemcpy(host.data(), h->tab, host.size() * 4, hipMemcpyDeviceToHost));
    for (int rel = -(len - 1); rel <= len - 1; ++rel) out[rel + len - 1] = host[(size_t)(rel + h->tab_len - 1)];
    return SVI_OK;
}

extern "C" svi_status svi_t5_create(const svi_t5_config* cfg, svi_t5** out) {
    SVI_REQUIRE(cfg && out, "svi_t5_create: null argument");
    SVI_REQUIRE(cfg->vocab > 0 && cfg->dim > 0 && cfg->dim % 8 == 0 && cfg->dim_attn > 0 && cfg->dim_attn % 8 == 0 && cfg->dim_ffn > 0 &&
                cfg->dim_ffn % 8 == 0 && cfg->num_heads > 0 && cfg->dim_attn % cfg->num_heads == 0 && cfg->num_layers > 0,
                "svi_t5_create: dim, dim_attn, dim_ffn must be positive multiples of 8 and heads must divide dim_attn");
    const int D = cfg->dim_attn / cfg->num_heads;
    SVI_REQUIRE(D == 32 || D == 64 || D == 80 || D == 128, "svi_t5_create: head dim %d (supported: 32, 64, 80, 128)", D);
    SVI_REQUIRE(cfg->num_buckets >= 4 && cfg->num_buckets % 2 == 0 && cfg->max_dist > cfg->num_buckets / 4, "svi_t5_create: bad bucket configuration");
    svi_t5* h = new (std::nothrow) svi_t5();
    if (!h) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    h->cfg = *cfg;
    *out = h;
    return SVI_OK;
}

extern "C" svi_status svi_t5_destroy(svi_t5* h) {
    if (!h) return SVI_OK;
    if (h->ws) (void)hipFree(h->ws);
    if (h->tab) (void)hipFree(h->tab);
    delete h;
    return SVI_OK;
}

namespace {
// expected shape of a WanTextEncoder state-dict key, or false
bool t5_expected(const svi_t5_config& c, const std::string& name, std::vector<int64_t>* shape) {
    if (name == "token_embedding.weight") { *shape = {c.vocab, c.dim}; return true; }
    if (name == "norm.weight") { *shape = {c.dim}; return true; }
    if (name == "pos_embedding.embedding.weight") { if (!c.shared_pos) return false; *shape = {c.num_buckets, c.num_heads}; return true; }
    if (name.rfind("blocks.", 0) != 0) return false;
    char* end = nullptr;
    const long i = strtol(name.c_str() + 7, &end, 10);
    if (end == name.c_str() + 7 || *end != '.' || i < 0 || i >= c.num_layers) return false;
    const std::string rest(end + 1);
    if (rest == "norm1.weight" || rest == "norm2.weight") { *shape = {c.dim}; return true; }
    if (rest == "attn.q.weight" || rest == "attn.k.weight" || rest == "attn.v.weight") { *shape = {c.dim_attn, c.dim}; return true; }
    if (rest == "attn.o.weight") { *shape = {c.dim, c.dim_attn}; return true; }
    if (rest == "ffn.gate.0.weight" || rest == "ffn.fc1.weight") { *shape = {c.dim_ffn, c.dim}; return true; }
    if (rest == "ffn.fc2.weight") { *shape = {c.dim, c.dim_ffn}; return true; }
    if (rest == "pos_embedding.embedding.weight") { if (c.shared_pos) return false; *shape = {c.num_buckets, c.num_heads}; return true; }
    return false;
}
std::vector<std::string> t5_keys(const svi_t5_config& c) {
    std::vector<std::string> k = {"token_embedding.weight", "norm.weight"};
    if (c.shared_pos) k.push_back("pos_embedding.embedding.weight");
    const char* per[] = {"norm1.weight", "norm2.weight", "attn.q.weight", "attn.k.weight", "attn.v.weight", "attn.o.weight",
                         "ffn.gate.0.weight", "ffn.fc1.weight", "ffn.fc2.weight"};
    for (int i = 0; i < c.num_layers; ++i) {
        for (const char* p : per) k.push_back("blocks." + std::to_string(i) + "." + p);
        if (!c.shared_pos) k.push_back("blocks." + std::to_string(i) + ".pos_embedding.embedding.weight");
    }
    return k;
}
}  // namespace

extern "C" svi_status svi_t5_bind_weight(svi_t5* h, const char* name, const void* dev_ptr, svi_dtype dtype, const int64_t* shape, int32_t rank) {
    SVI_REQUIRE(h && name && dev_ptr && shape && rank >= 1 && rank <= 2, "svi_t5_bind_weight: bad argument");
    SVI_REQUIRE(dtype == SVI_BF16, "text encoder parameter '%s' must be bf16 (the pipeline keeps this module in bf16)", name);
    SVI_REQUIRE(((uintptr_t)dev_ptr % 16) == 0, "text encoder parameter '%s' is not 16-byte aligned", name);
    std::vector<int64_t> want;
    if (!t5_expected(h->cfg, name, &want)) { svi_set_error("unknown text encoder parameter '%s'", name); return SVI_ERR_INVALID; }
    bool ok = (size_t)rank == want.size();
    for (int i = 0; ok && i < rank; ++i) ok = shape[i] == want[i];
    if (!ok) { svi_set_error("shape mismatch for text encoder parameter '%s'", name); return SVI_ERR_INVALID; }
    Param q; q.p = dev_ptr; q.shape = want;
    h->w[name] = q;
    return SVI_OK;
}

extern "C" svi_status svi_t5_check_bound(svi_t5* h) {
    SVI_REQUIRE(h, "null handle");
    for (const auto& k : t5_keys(h->cfg))
        if (!h->w.count(k)) { svi_set_error("text encoder parameter '%s' was never bound", k.c_str()); return SVI_ERR_UNBOUND; }
    return SVI_OK;
}

// ids int64 [L] on the device; keys = the first n_valid positions (the mask of the reference is a prefix mask: tokenizer padding);
// query rows 0..rows-1 are computed (n_valid <= rows <= L), the remaining rows of out [L, dim] bf16 are zero.
extern "C" svi_status svi_t5_forward(svi_t5* h, const int64_t* ids, int32_t L, int32_t n_valid, int32_t rows, void* out, svi_stream stream) {
    SVI_REQUIRE(h && ids && out, "svi_t5_forward: null argument");
    SVI_REQUIRE(L > 0 && n_valid >= 1 && n_valid <= rows && rows <= L && L <= 2048, "svi_t5_forward: need 1 <= n_valid <= rows <= L <= 2048 (got %d, %d, %d)",
                n_valid, rows, L);
    SVI_REQUIRE_DEVICE(h);
    SVI_TRY(svi_t5_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const svi_t5_config& c = h->cfg;
    const int R = rows, dim = c.dim, da = c.dim_attn, df = c.dim_ffn, D = da / c.num_heads;
    SVI_TRY(t5_device_table(h, L));
    auto al = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t sx = al((size_t)R * dim * 2), sa = al((size_t)R * da * 2), sf = al((size_t)R * df * 2);
    SVI_TRY(grow(&h->ws, &h->ws_bytes, 2 * sx + 4 * sa + 2 * sf, "text encoder workspace"));
    char* p = h->ws;
    bf16* X = reinterpret_cast<bf16*>(p); p += sx;
    bf16* N = reinterpret_cast<bf16*>(p); p += sx;
    bf16* Qb = reinterpret_cast<bf16*>(p); p += sa;
    bf16* Kb = reinterpret_cast<bf16*>(p); p += sa;
    bf16* Vb = reinterpret_cast<bf16*>(p); p += sa;
    bf16* Ab = reinterpret_cast<bf16*>(p); p += sa;
    bf16* G = reinterpret_cast<bf16*>(p); p += sf;
    bf16* F1 = reinterpret_cast<bf16*>(p);
    auto W = [&](const std::string& k) { return reinterpret_cast<const bf16*>(h->w[k].p); };
    auto linear = [&](const bf16* A, int K, const bf16* Wm, int Nn, bf16* Cm, const bf16* res) {
        SviGemmArgs g{};
        g.A = A; g.lda = K; g.W = Wm; g.ldw = K; g.C = Cm; g.ldc = Nn; g.M = R; g.N = Nn; g.K = K;
        g.epi = res ? SVI_EPI_BIAS_GATE_RES : SVI_EPI_BIAS;               // res: bf16(res + bf16(acc)), the module output rounded first (t5:137-138)
        g.res = res; g.ldres = Nn;
        g.skinny = 1;                                                      // taken when R <= 128 (a prompt's valid tokens), see svi_gemm.hip
        return svi_launch_gemm(g, st);
    };
    auto norm = [&](const bf16* in, bf16* o, const bf16* w) {
        hipLaunchKernelGGL(t5_norm_kernel, dim3((R + 3) / 4), dim3(256), 0, st, in, o, w, R, dim, 1e-6f);
    };
    hipLaunchKernelGGL(t5_embed_kernel, dim3(R), dim3(256), 0, st, ids, W("token_embedding.weight"), X, R, dim, c.vocab);
    SVI_LAUNCH_CHECK();
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string b = "blocks." + std::to_string(i) + ".";
        norm(X, N, W(b + "norm1.weight"));
        SVI_LAUNCH_CHECK();
        SVI_TRY(linear(N, dim, W(b + "attn.q.weight"), da, Qb, nullptr));
        SVI_TRY(linear(N, dim, W(b + "attn.k.weight"), da, Kb, nullptr));
        SVI_TRY(linear(N, dim, W(b + "attn.v.weight"), da, Vb, nullptr));
        const bf16* emb = W(c.shared_pos ? std::string("pos_embedding.embedding.weight") : b + "pos_embedding.embedding.weight");
        SVI_TRY((launch_enc_attention<bf16, true>(Qb, da, Kb, da, Vb, da, Ab, da, R, n_valid, c.num_heads, D, 1.0f, emb, h->tab,
                                                  h->tab_len - 1, st)));
        SVI_TRY(linear(Ab, da, W(b + "attn.o.weight"), dim, X, X));
        norm(X, N, W(b + "norm2.weight"));
        SVI_LAUNCH_CHECK();
        SVI_TRY(linear(N, dim, W(b + "ffn.gate.0.weight"), df, G, nullptr));
        SVI_TRY(linear(N, dim, W(b + "ffn.fc1.weight"), df, F1, nullptr));
        const long n = (long)R * df;
        hipLaunchKernelGGL(t5_gated_gelu_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, G, F1, G, n);
        SVI_LAUNCH_CHECK();
        SVI_TRY(linear(G, df, W(b + "ffn.fc2.weight"), dim, X, X));
    }
    hipLaunchKernelGGL(t5_norm_kernel, dim3((R + 3) / 4), dim3(256), 0, st, X, reinterpret_cast<bf16*>(out), W("norm.weight"), R, dim, 1e-6f);
    if (R < L) {
        const long n = (long)(L - R) * dim;
        hipLaunchKernelGGL(zero_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<bf16*>(out) + (size_t)R * dim, n);
    }
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ================================================================= CLIP ============================================================
struct svi_clip {
    int device = -1;
    svi_clip_config cfg{};
    std::map<std::string, Param> w;
    char* ws = nullptr; size_t ws_bytes = 0;
};

namespace {
bool clip_expected(const svi_clip_config& c, const std::string& name, std::vector<int64_t>* shape, bool* used) {
    const int np = (c.image_size / c.patch_size) * (c.image_size / c.patch_size), md = c.dim * c.mlp_ratio;
    *used = true;
    if (name == "patch_embedding.weight") { *shape = {c.dim, 3, c.patch_size, c.patch_size}; return true; }
    if (name == "cls_embedding") { *shape = {1, 1, c.dim}; return true; }
    if (name == "pos_embedding") { *shape = {1, np + 1, c.dim}; return true; }
    if (name == "pre_norm.weight" || name == "pre_norm.bias") { *shape = {c.dim}; return true; }
    if (name == "post_norm.weight" || name == "post_norm.bias" || name == "head") { *used = false; shape->clear(); return true; }   // not on encode_image's path
    if (name.rfind("transformer.", 0) != 0) return false;
    char* end = nullptr;
    const long i = strtol(name.c_str() + 12, &end, 10);
    if (end == name.c_str() + 12 || *end != '.' || i < 0 || i >= c.num_layers) return false;
    if (i >= c.layers_used) *used = false;
    const std::string rest(end + 1);
    if (rest == "norm1.weight" || rest == "norm1.bias" || rest == "norm2.weight" || rest == "norm2.bias" || rest == "attn.proj.bias" ||
        rest == "mlp.2.bias") { *shape = {c.dim}; return true; }
    if (rest == "attn.to_qkv.weight") { *shape = {3 * c.dim, c.dim}; return true; }
    if (rest == "attn.to_qkv.bias") { *shape = {3 * c.dim}; return true; }
    if (rest == "attn.proj.weight") { *shape = {c.dim, c.dim}; return true; }
    if (rest == "mlp.0.weight") { *shape = {md, c.dim}; return true; }
    if (rest == "mlp.0.bias") { *shape = {md}; return true; }
    if (rest == "mlp.2.weight") { *shape = {c.dim, md}; return true; }
    return false;
}
std::vector<std::string> clip_keys(const svi_clip_config& c) {
    std::vector<std::string> k = {"patch_embedding.weight", "cls_embedding", "pos_embedding", "pre_norm.weight", "pre_norm.bias"};
    const char* per[] = {"norm1.weight", "norm1.bias", "attn.to_qkv.weight", "attn.to_qkv.bias", "attn.proj.weight", "attn.proj.bias",
                         "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"};
    for (int i = 0; i < c.layers_used; ++i)
        for (const char* p : per) k.push_back("transformer." + std::to_string(i) + "." + p);
    return k;
}
}  // namespace

extern "C" svi_status svi_clip_create(const svi_clip_config* cfg, svi_clip** out) {
    SVI_REQUIRE(cfg && out, "svi_clip_create: null argument");
    SVI_REQUIRE(cfg->image_size > 0 && cfg->patch_size > 0 && cfg->image_size % cfg->patch_size == 0 && cfg->dim > 0 && cfg->dim % 4 == 0 &&
                cfg->num_heads > 0 && cfg->dim % cfg->num_heads == 0 && cfg->mlp_ratio > 0 && cfg->num_layers > 0 && cfg->layers_used > 0 &&
                cfg->layers_used <= cfg->num_layers && cfg->norm_eps > 0.f,
                "svi_clip_create: bad configuration (patch must divide image size, heads must divide dim, dim %% 4 == 0)");
    SVI_REQUIRE((3 * cfg->patch_size * cfg->patch_size) % 4 == 0, "svi_clip_create: 3 * patch_size^2 must be a multiple of 4");
    const int D = cfg->dim / cfg->num_heads;
    SVI_REQUIRE(D == 32 || D == 64 || D == 80 || D == 128, "svi_clip_create: head dim %d (supported: 32, 64, 80, 128)", D);
    svi_clip* h = new (std::nothrow) svi_clip();
    if (!h) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    h->cfg = *cfg;
    *out = h;
    return SVI_OK;
}

extern "C" svi_status svi_clip_destroy(svi_clip* h) {
    if (!h) return SVI_OK;
    if (h->ws) (void)hipFree(h->ws);
    delete h;
    return SVI_OK;
}

// name = VisionTransformer state-dict key (the "model.visual." prefix of WanImageEncoder stripped by the caller); fp32.
extern "C" svi_status svi_clip_bind_weight(svi_clip* h, const char* name, const void* dev_ptr, svi_dtype dtype, const int64_t* shape, int32_t rank) {
    SVI_REQUIRE(h && name && dev_ptr && shape && rank >= 1 && rank <= 4, "svi_clip_bind_weight: bad argument");
    SVI_REQUIRE(dtype == SVI_F32, "image encoder parameter '%s' must be fp32 (SVI runs this module in fp32)", name);
    SVI_REQUIRE(((uintptr_t)dev_ptr % 16) == 0, "image encoder parameter '%s' is not 16-byte aligned", name);
    std::vector<int64_t> want;
    bool used = true;
    if (!clip_expected(h->cfg, name, &want, &used)) { svi_set_error("unknown image encoder parameter '%s'", name); return SVI_ERR_INVALID; }
    if (!used) return SVI_OK;
    bool ok = (size_t)rank == want.size();
    for (int i = 0; ok && i < rank; ++i) ok = shape[i] == want[i];
    if (!ok) { svi_set_error("shape mismatch for image encoder parameter '%s'", name); return SVI_ERR_INVALID; }
    Param q; q.p = dev_ptr; q.shape = want;
    h->w[name] = q;
    return SVI_OK;
}

extern "C" svi_status svi_clip_check_bound(svi_clip* h) {
    SVI_REQUIRE(h, "null handle");
    for (const auto& k : clip_keys(h->cfg))
        if (!h->w.count(k)) { svi_set_error("image encoder parameter '%s' was never bound", k.c_str()); return SVI_ERR_UNBOUND; }
    return SVI_OK;
}

extern "C" svi_status svi_clip_tokens(svi_clip* h, int32_t* tokens, int32_t* dim) {
    SVI_REQUIRE(h && tokens && dim, "svi_clip_tokens: null argument");
    const int np = h->cfg.image_size / h->cfg.patch_size;
    *tokens = np * np + 1; *dim = h->cfg.dim;
    return SVI_OK;
}

// images f32 [B, 3, H, W] in [-1, 1]  ->  out f32 [B, tokens, dim]: the hidden states after block layers_used-1
extern "C" svi_status svi_clip_encode_image(svi_clip* h, const float* images, int32_t B, int32_t H, int32_t Wd, float* out, svi_stream stream) {
    SVI_REQUIRE(h && images && out && B > 0 && H > 0 && Wd > 0, "svi_clip_encode_image: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_TRY(svi_clip_check_bound(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const svi_clip_config& c = h->cfg;
    const int S = c.image_size, ps = c.patch_size, npx = S / ps, np = npx * npx, T = np + 1, dim = c.dim, md = dim * c.mlp_ratio, D = dim / c.num_heads;
    const int kp = 3 * ps * ps;
    auto al = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t s_patch = al((size_t)np * kp * 4), s_x = al((size_t)T * dim * 4), s_qkv = al((size_t)T * 3 * dim * 4), s_h = al((size_t)T * md * 4);
    SVI_TRY(grow(&h->ws, &h->ws_bytes, s_patch + 3 * s_x + s_qkv + s_h, "image encoder workspace"));
    char* p = h->ws;
    float* P = reinterpret_cast<float*>(p); p += s_patch;
    float* X0 = reinterpret_cast<float*>(p); p += s_x;
    float* N = reinterpret_cast<float*>(p); p += s_x;
    float* A = reinterpret_cast<float*>(p); p += s_x;
    float* QKV = reinterpret_cast<float*>(p); p += s_qkv;
    float* Hd = reinterpret_cast<float*>(p);
    auto W = [&](const std::string& k) { return reinterpret_cast<const float*>(h->w[k].p); };
    auto ln = [&](const float* in, float* o, const std::string& name) {
        hipLaunchKernelGGL(layernorm_f32_kernel, dim3((T + 3) / 4), dim3(256), 0, st, in, o, W(name + ".weight"), W(name + ".bias"), T, dim, c.norm_eps);
    };
    const float scale = 1.0f / sqrtf((float)D);
    for (int b = 0; b < B; ++b) {
        float* X = out + (size_t)b * T * dim;                        // the residual stream lives in the output
        hipLaunchKernelGGL(clip_preprocess_kernel, dim3((3 * S * S + 255) / 256), dim3(256), 0, st, images + (size_t)b * 3 * H * Wd, H, Wd, P, S, ps, kp);
        hipLaunchKernelGGL(clip_cls_kernel, dim3((dim + 255) / 256), dim3(256), 0, st, W("cls_embedding"), W("pos_embedding"), X0, dim);
        SVI_LAUNCH_CHECK();
        // patch embedding (bias-free Conv2d, stride = kernel) + positional embedding rows 1..np as the residual operand
        SVI_TRY(svi_launch_gemm_f32(P, kp, W("patch_embedding.weight"), kp, nullptr, X0 + dim, dim, np, dim, kp, W("pos_embedding") + dim, dim, st));
        ln(X0, X, "pre_norm");
        SVI_LAUNCH_CHECK();
        for (int i = 0; i < c.layers_used; ++i) {
            const std::string t = "transformer." + std::to_string(i) + ".";
            ln(X, N, t + "norm1");
            SVI_LAUNCH_CHECK();
            SVI_TRY(svi_launch_gemm_f32(N, dim, W(t + "attn.to_qkv.weight"), dim, W(t + "attn.to_qkv.bias"), QKV, 3 * dim, T, 3 * dim, dim, nullptr, 0, st));
            SVI_TRY((launch_enc_attention<float, false>(QKV, 3 * dim, QKV + dim, 3 * dim, QKV + 2 * dim, 3 * dim, A, dim, T, T, c.num_heads, D, scale,
                                                        nullptr, nullptr, 0, st)));
            SVI_TRY(svi_launch_gemm_f32(A, dim, W(t + "attn.proj.weight"), dim, W(t + "attn.proj.bias"), X, dim, T, dim, dim, X, dim, st));
            ln(X, N, t + "norm2");
            SVI_LAUNCH_CHECK();
            SVI_TRY(svi_launch_gemm_f32(N, dim, W(t + "mlp.0.weight"), dim, W(t + "mlp.0.bias"), Hd, md, T, md, dim, nullptr, 0, st));
            const long n = (long)T * md;
            hipLaunchKernelGGL(gelu_erf_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Hd, n);
            SVI_LAUNCH_CHECK();
            SVI_TRY(svi_launch_gemm_f32(Hd, md, W(t + "mlp.2.weight"), md, W(t + "mlp.2.bias"), X, dim, T, dim, md, X, dim, st));
        }
    }
    return SVI_OK;
}
