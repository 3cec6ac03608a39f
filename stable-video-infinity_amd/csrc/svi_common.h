// svi_common.h — shared device helpers and host-side error plumbing for libsvi_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/svi_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SVI_WAVE 64

// ---- host: status + thread-local message -------------------------------------------------
void svi_set_error(const char* fmt, ...);

#define SVI_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            svi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SVI_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define SVI_REQUIRE(cond, ...)                                                                \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            svi_set_error(__VA_ARGS__);                                                       \
            return SVI_ERR_INVALID;                                                           \
        }                                                                                     \
    } while (0)

#define SVI_LAUNCH_CHECK()                                                                    \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            svi_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return SVI_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define SVI_TRY(expr)                                                                         \
    do {                                                                                      \
        svi_status _s = (expr);                                                               \
        if (_s != SVI_OK) return _s;                                                          \
    } while (0)

// ---- process-wide switches, read ONCE (first use) ------------------------------------------------
// A/B aids that select between kernels computing the SAME result (bit-identical or within the stated parity bounds); they
// never change what is computed.  Switches that make results wrong (timing ablations) exist only in variant builds made with
// -DSVI_ABLATIONS (tools/build_variant.py) and are absent from the product library.
struct SviSwitches {
    int flash_kernel = 0;        // SVI_FLASH_KERNEL = 1 | 2 : force flash_fwd_kernel / flash_fwd2_kernel (0: by key count)
    int gemm_kernel = 0;         // SVI_GEMM_KERNEL = 128 | 192 | 259 | 260 : force the 128^2 kernel / the 256 x 192 tile / the 256^2 tile with four / two phases per K tile (0: by tile count)
    int gemm_pf = 0;             // SVI_GEMM_PF = 1 | 4 | 8 : the 128^2 kernel's loop — 1 = one K tile of loads in flight, 4 = four, 8 = four on eight waves (0: 8 when the launch is at most one workgroup per CU, else 1)
    int gemm_gm = 0;             // SVI_GEMM_GM = n >= 1 : row panels per tile group (0: per shape)
    int vae_exact_fp32 = 0;      // SVI_VAE_EXACT_FP32 : fp32-MFMA convolution everywhere
    int vae_no_x2h = 0;          // SVI_VAE_X2H = 0 : the three-term bf16 convolution also where the two-term fp16 form applies (same parity bounds)
    int vae_dma = 1;             // SVI_VAE_DMA = 0 : residual-block convolutions split their fp32 input on the fly (conv_igemm_x3_kernel<true>) instead of reading
                                 // the fp16 word pairs the producing RMS_norm wrote, staged by LDS-DMA (conv_dma2h_kernel); bit-identical results
    int vae_pair = 1;            // SVI_VAE_PAIR = 0 : the residual-block convolutions one 256-pixel tile per workgroup (conv_dma2h_kernel<3>) instead of two tiles that share
                                 // every K step's weights (conv_dma2h_pair_kernel: 26 % fewer bytes through the LDS fill path per MFMA; bit-identical)
    int vae_tile_order = 1;      // SVI_VAE_TILE_ORDER = 0 : the plane-fed convolution's tiles in pixel order (one frame after the other) instead of groups of ~16 image
                                 // rows walked through all frames (same results bit for bit; which order finds its earlier frames' rows still cached)
    int vae_up_phases = 1;       // SVI_VAE_UP_PHASES = 0 : the convolution behind a nearest x2 upsample as ONE 3x3 convolution reading through the upsample (9 taps)
                                 // instead of four 2x2 convolutions of the small image with pre-summed kernels (4 taps; same sum up to fp32 rounding of the weights)
    int flash_two_pass = 1;      // SVI_FLASH_TWO_PASS = 0 : the long-sequence attention as ONE complete pass (tracked maximum) instead of the
                                 // optimistic pass + flagged second pass (same result within the attention tolerance; bit-identical on benign operands)
    int flash_m16 = 1;           // SVI_FLASH_M16 = 0 : the optimistic attention pass on v_mfma_f32_32x32x16_bf16 (flash_fwd2_kernel, rounds 2-6) instead of 16x16x32
                                 // (flash_fwd3_kernel: the shape the part's power limit favours); same softmax, results equal within the attention tolerance (the
                                 // matrix instruction sums a row's 128 channels in another order)
    int flash_split = 0;         // SVI_FLASH_SPLIT = 1 : never cut the key axis of the long-sequence attention (bit-identical to the unsplit kernel); 2..4: that many
                                 // pieces wherever the key axis allows; 0 (default): where the workgroup count fills the chip's last round poorly (svi_attention.hip)
    int rms_rows = 1;            // SVI_RMS_ROWS = 0 : RMSNorm (+RoPE) and LayerNorm (+modulate) with one row per wave (the generic kernels) also for the DiT's shapes, instead of four rows per wave
                                 // with the next row requested ahead and the gain vector kept in registers (bit-identical)
    int qk_fused = 1;            // SVI_QK_FUSED = 0 : the self-attention q and k projections as two launches instead of one N = 2 dim launch over the two weight
                                 // matrices (bit-identical: per element the same kernel and the same k order)
    int mx8_fused = 1;           // SVI_MX8_FUSED = 0 : (opt-in MX-fp8 MLP) ffn1 stores its bf16 result and a separate launch quantises it, instead of quantising in
                                 // ffn1's epilogue (bit-identical; saves one write and one read of the [L, ffn_dim] activation)
    int attn_qk8 = 0;            // SVI_ATTN_QK8 = 1 : (opt-in, never the default) the long-sequence attention quantises Q and K to MX e4m3 (one E8M0 scale per 32
                                 // channels) and takes QK^T on the scaled fp8 MFMA at twice the bf16 rate; P·V stays bf16.  Arithmetic the reference never
                                 // performs (its dispatch accepts a quantised-QK^T backend, wan_video_dit.py:116-147): own oracle, own tolerance, own bench line
    int qk8_fused = 1;           // SVI_QK8_FUSED = 0 : (opt-in fp8 QK^T) the DiT's RMSNorm + RoPE launch writes bf16 q | k and the attention call quantises them with two more
                                 // launches, instead of the RMSNorm + RoPE kernel writing the e4m3 rows and block scales itself (bit-identical)
    int cross_fused = 1;         // SVI_CROSS_FUSED = 0 : the cross-attention query is normalised by its own kernel (RMSNorm in place) and attended by flash_fwd_kernel<1>
                                 // (rounds 1-4); default: the q projection's epilogue leaves the row sums of squares and flash_cross_kernel normalises as it reads
    int cross_dedup = 1;         // SVI_CROSS_DEDUP = 0 : cross-attention walks every context row even where the prompt embedding's trailing rows are
                                 // identical (the prompter's zero padding); default: m identical keys = one key counted m times (same softmax)
    int t5_host_buckets = 0;     // SVI_T5_BUCKETS = host : the text encoder's relative-position bucket table in the HOST's fp32 arithmetic (what the
                                 // reference module computes on a CPU — the arithmetic the committed fixtures were made with) instead of the device's
    int ws_limit_mb = 0;         // SVI_WS_LIMIT_MB = n : a DiT workspace beyond n MiB is refused as if the allocation had failed (SVI_ERR_OOM): a budget for callers who
                                 // share the device, and the way tests reach the stacked CFG pair's out-of-memory fall-back without exhausting a 288 GB part
#ifdef SVI_ABLATIONS
    int flash_abl = 0, gemm_epi_abl = 0, vae_abl = 0, flash_assume_prescaled = 0;
#endif
};
const SviSwitches& svi_switches();

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: raise it once per (device, kernel).
svi_status svi_ensure_lds(const void* kernel, int bytes);
// Library-owned device buffers outside any handle, one per (device, stream, kind); see svi_api.hip.  `user_out`: a host-side word
// that lives with the buffer (only the thread driving that stream touches it).
enum SviBufKind { SVI_BUF_FLASH_FLAGS = 0, SVI_BUF_SEAM_SCRATCH = 1, SVI_BUF_SEAM_SMALL = 2, SVI_BUF_FLASH_SPLIT = 3, SVI_BUF_FLASH_QK8 = 4 };
svi_status svi_stream_buffer(int kind, hipStream_t st, size_t bytes, void** out, long** user_out);
// The device current on this thread, or -1 (message set).
int svi_current_device();
// Handles belong to ONE device (workspace, packed weights, LDS attributes live there): the first compute call claims the device
// current on the calling thread, later calls under another current device are refused instead of faulting.
svi_status svi_claim_device(int* handle_device);
#define SVI_REQUIRE_DEVICE(h) SVI_TRY(svi_claim_device(&(h)->device))

// ---- device helpers ----------------------------------------------------------------------
#ifdef __HIPCC__
// value of a float after a round trip through bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32):
// the reference materialises a bf16 tensor after every op, this restates that rounding point.
__device__ __forceinline__ float rbf(float v) { return (float)(bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st_bf16x8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

// Eight consecutive elements of a row (a quarter of a 32-element MX block) -> e4m3 bytes + the block's E8M0 code.  Lane layout: lane & 3 = the
// quarter inside its block, lane & 15 = the eighth inside its 128-element group (whose four codes make one dword, written by the group's first
// lane); the 16 lanes of a group hold the same row and are all live or all dead.  Shared by mx8_quantize_kernel and the GEMM epilogue that
// quantises its own result (SviGemmArgs::q8): the same operations, the same bits.
__device__ __forceinline__ void mx8_quant8(const float (&v)[8], bool live, int row, int c8, unsigned char* __restrict__ q, int ldq,
                                           unsigned* __restrict__ scales, int sc_rows) {
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    const int eb = (int)((__float_as_uint(amax) >> 23) & 0xffu);            // biased exponent of the block maximum (0 for zero / subnormal)
    const int E = max(eb - 8, 0);                                           // E8M0 code of the shared scale 2^(E - 127)
    const float inv = __uint_as_float((unsigned)(254 - E) << 23);           // 2^(127 - E), exact
    unsigned w[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = fminf(fmaxf(v[4 * h2 + e] * inv, -448.f), 448.f);      // saturate to the e4m3 range, then round to nearest even
        unsigned pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], pk, true);
        w[h2] = pk;
    }
    if (live) *reinterpret_cast<u32x2*>(q + (size_t)row * ldq + c8 * 8) = u32x2{w[0], w[1]};
    // the four block codes of a 128-element group -> one dword, written by the group's first lane
    const int lane = threadIdx.x & 63;
    const unsigned e0 = (unsigned)E;
    const unsigned e1 = (unsigned)__shfl(E, (lane & ~15) + 4), e2 = (unsigned)__shfl(E, (lane & ~15) + 8), e3 = (unsigned)__shfl(E, (lane & ~15) + 12);
    if (live && (lane & 15) == 0) scales[(size_t)(c8 >> 4) * sc_rows + row] = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
}

// gelu_tanh(x) = 0.5 x (1 + tanh(u)), u = sqrt(2/pi)(x + 0.044715 x^3).  1 + tanh(u) = 2 sigmoid(2u), so
// gelu_tanh(x) = x / (1 + exp(-2u)): one v_exp + one v_rcp instead of a tanhf expansion (fp32-accurate;
// the result is rounded to bf16 by the caller).
// With the constants folded: 2^(-2 log2(e) u) = 2^(x (A x^2 + B)), A = -2 log2(e) k0 k1, B = -2 log2(e) k0 (one multiply fewer; every kernel
// uses this one function, so a shard on one tile size and the whole on another round alike).
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float A = -2.0f * 1.4426950408889634f * 0.7978845608028654f * 0.044715f, B = -2.0f * 1.4426950408889634f * 0.7978845608028654f;
    const float a = x * __builtin_fmaf(x * x, A, B);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
// Tried: two elements at a time on the packed fp32 pipe (v_pk_mul / v_pk_fma / v_pk_add around the scalar v_exp / v_rcp) in the 256-row
// tiles' epilogue — the same bits, and no faster (ffn1 834-860 us either way, tools/gemm_ab.py on one box): the two transcendentals per element,
// not the five multiply-adds, are what the GELU epilogue costs.
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
#endif

// ---- event profiler (svi_prof_enable / svi_prof_summary) ---------------------------------------
enum SviProfTag {
    PROF_LN = 0, PROF_GEMM_QKV, PROF_RMS_ROPE, PROF_FLASH_SELF, PROF_GEMM_O, PROF_GEMM_CROSS, PROF_FLASH_CROSS,
    PROF_GEMM_FFN1, PROF_GEMM_FFN2, PROF_EMBED, PROF_HEAD, PROF_VAE_CONV, PROF_VAE_OTHER, PROF_NTAGS
};
extern bool g_svi_prof_on;
extern unsigned g_svi_prof_mask;             // bit t: tag t is recorded (svi_prof_select; all tags by default)
void svi_prof_begin_impl(int tag, hipStream_t st);
void svi_prof_end_impl(int tag, hipStream_t st);
struct SviProfScope {
    int tag; hipStream_t st; bool on;
    SviProfScope(int t, hipStream_t s) : tag(t), st(s), on(g_svi_prof_on && ((g_svi_prof_mask >> t) & 1u)) { if (on) svi_prof_begin_impl(tag, st); }
    ~SviProfScope() { if (on) svi_prof_end_impl(tag, st); }
};

// ---- kernel launchers shared between translation units (all enqueue on `st`) ----------------
struct SviGemmArgs {
    const bf16* A; int lda;
    const bf16* W; int ldw;
    bf16* C; int ldc;
    int M, N, K;
    const bf16* bias; int bias_along_m;
    int epi;
    const float* gate;
    const bf16* res; int ldres;
    int sel_m, sel_n;           // > 0: choose the kernel as for a problem with this many rows / columns (stacked samples keep the per-sample choice,
                                // so every row sees the same kernel — and the same bits — as in a per-sample launch); 0: by M / N
    int skinny;                 // != 0: rows <= 128 against a large weight matrix (the text encoder): the weight-streaming kernel
    // svi_launch_gemm_mx8 only: when q8 is set the epilogue's bf16 result is not stored but quantised in place to MX e4m3 ([M][ldq8] bytes) with its
    // E8M0 block scales ([N/128][q8_sc_rows] dwords) — bit for bit what svi_launch_mx8_quantize makes of the stored bf16 tensor.  N % 256 == 0.
    unsigned char* q8; int ldq8; unsigned* q8s; int q8_sc_rows;
    // two weight matrices side by side along N (the self-attention q | k projections as ONE launch, wan_video_dit.py:227-228): columns n >= n_split
    // take row n - n_split of W2 (same ldw) and bias2.  n_split must be a multiple of the tile width of the kernel that runs (the launcher falls back
    // to two launches otherwise); the weights stay the caller's tensors — nothing is packed, so an in-place LoRA merge needs no re-bind.
    const bf16* W2; const bf16* bias2; int n_split;
    // row statistics of the OUTPUT (the cross-attention q projection feeding an RMSNorm over the full width, dit:194-197,296): when set, every tiled
    // kernel's epilogue also leaves the sum of squares of the rounded bf16 results of row m over each aligned 64-column group g in rowss[g * ldss + m]
    // (N % 64 == 0, SVI_EPI_BIAS).  Per group: a lane sums its 8 columns in order, then a fixed xor-1/2/4 tree over the group's 8 lanes — the same tree in
    // the 128^2, 256 x 192 and 256^2 kernels, so the statistics do not depend on which kernel (or how many rows) a launch ran with.
    float* rowss; int ldss;
};
unsigned long long svi_stream_buffer_generation();       // moves whenever a per-stream library buffer is freed (svi_api.hip)
svi_status svi_launch_gemm(const SviGemmArgs& g, hipStream_t st);
int svi_gemm_choose(const SviGemmArgs& g, int compute_units);          // which kernel svi_launch_gemm takes (pure; see svi_gemm.hip)
// how svi_launch_flash splits a launch into work items (pure; see svi_attention.hip): items [0, whole) run whole, the rest in `pieces` workgroups each
struct SviFlashSplit { int whole, pieces, qblocks, heads; };
SviFlashSplit svi_flash_plan(int Lq, int Lk, int heads, int cus, int* kernel_out);
// MX-fp8 (opt-in): bf16 -> e4m3 + E8M0 block scales ([K/128][sc_rows] dwords), and C = epi(A8 W8^T) with g.A / g.W e4m3, lda / ldw in bytes
svi_status svi_launch_mx8_quantize(const bf16* x, int ldx, int rows, int K, unsigned char* q, int ldq, unsigned* scales, int sc_rows, hipStream_t st);
svi_status svi_launch_gemm_mx8(const SviGemmArgs& g, const unsigned* a_scales, int sc_rows, hipStream_t st);
// the same with the block scales on the W operand's rows and unit scales on A (the transposed value projection; bias epilogue only)
svi_status svi_launch_gemm_mx8_wscaled(const SviGemmArgs& g, const unsigned* w_scales, int sc_rows, hipStream_t st);

// Q [Lq, ldq], K [Lk, ldk] token-major with head hd at column hd*128; VT [(n*128), ldvt] = V transposed
// (row = channel, col = key; columns >= Lk up to the next multiple of 8 must be readable and finite).
// q_prescaled != 0: Q already carries softmax_scale * log2(e) = 1.4426950408889634 / sqrt(128)  (SVI_QK_SCALE_LOG2E).
#define SVI_QK_SCALE_LOG2E 0.12751743f
// (svi_launch_flash: q/k token-major with row strides, V TRANSPOSED [heads*128, ldvt])
// key_tail (device, optional; short key axes only): {n, m} — attend to keys 0 .. n-1 and count key n-1 m times (keys n-1 .. Lk-1 are identical)
// qk8 (opt-in fp8 QK^T only): Q and K already quantised by the caller into the buffers svi_flash_qk8_prepare handed out (the DiT's RMSNorm + RoPE kernel
// writes them instead of the bf16 rows); nullptr: svi_launch_flash quantises the bf16 operands itself when the mode is on.
struct SviQk8 { unsigned char* q8; unsigned char* k8; unsigned* qs; unsigned* ks; int ld8, qs_rows, ks_rows; };
svi_status svi_launch_flash(const bf16* Q, int ldq, const bf16* K, int ldk, const bf16* VT, int ldvt,
                            bf16* O, int ldo, int Lq, int Lk, int num_heads, int q_prescaled, hipStream_t st, const int* key_tail = nullptr, const SviQk8* qk8 = nullptr);
// Cross-attention over a SHORT key axis (the prompt: <= 512 keys, a few dozen distinct) with the query's RMSNorm applied as the rows are read:
//   q' = bf16(bf16(bf16(q * rs[row]) * gain) * out_scale)      — RMSNorm.forward's rounding points (dit:192-197) and the scale folded in as svi_launch_rmsnorm_rope does
// Q holds the RAW projection output; rs[row] = rsqrt(mean(q^2) + eps) from svi_launch_row_rs; norm == nullptr: Q is used as it is (pre-scaled).
// One [Lq, D] read and one write per launch instead of the separate normalisation's extra round trip.
struct SviQNorm { const float* rs; const bf16* gain; float out_scale; };
// key_blocks: the caller's HOST copy of ceil(key_tail[0] / 32) — how many 32-key blocks are walked; 1 .. 4: K / V^T stay resident in LDS (kernel instantiated
// per block count), more: the streaming kernel; <= 0: not known on the host -> the streaming kernel (any count).  Ignored without a key_tail (Lk decides).
svi_status svi_launch_flash_cross(const bf16* Q, int ldq, const bf16* K, int ldk, const bf16* VT, int ldvt, bf16* O, int ldo, int Lq, int Lk, int num_heads,
                                  hipStream_t st, const int* key_tail, const SviQNorm* norm, int key_blocks = 0);
// rs[m] = rsqrt(sum_g rowss[g * ldss + m] / dim + eps), groups summed in index order (rowss: see SviGemmArgs)
svi_status svi_launch_row_rs(const float* rowss, int groups, int ldss, int rows, int dim, float eps, float* rs, hipStream_t st);
// *use = whether svi_launch_flash will run this shape on the fp8 QK^T kernel (switch on, long key axis); if so `out` names the per-stream operand buffers
// batch > 1: room for that many samples stacked one under the other (sample s: rows [s Lq, (s+1) Lq) of q8 / qs, [s Lk, ..) of k8 / ks; svi_qk8_sample)
svi_status svi_flash_qk8_prepare(int Lq, int Lk, int num_heads, hipStream_t st, SviQk8* out, bool* use, int batch = 1);
inline SviQk8 svi_qk8_sample(const SviQk8& b, int s, int Lq, int Lk) {
    SviQk8 v = b;
    v.q8 += (size_t)s * Lq * b.ld8; v.k8 += (size_t)s * Lk * b.ld8; v.qs += (size_t)s * Lq; v.ks += (size_t)s * Lk;
    return v;
}

svi_status svi_launch_ln_mod(const bf16* x, int ldx, bf16* out, int ldo, int rows, int dim, float eps,
                             const bf16* w, const bf16* b, const float* shift, const float* scale1p,
                             hipStream_t st);
struct SviRope {                // device tables of (cos,sin) pairs, fp32
    const float2* tab_f; const float2* tab_h; const float2* tab_w;
    int npf, nph, npw;          // complex pairs per head owned by the frame / height / width axis
    int f, h, w;
    int row0;                   // token index of row 0 of the launch (sequence-parallel shards start mid-grid)
    int period;                 // > 0: rows are `rows / period` samples of `period` rows stacked one under the other; token = row0 + row % period
                                // (the stacked CFG pair: period = L on one rank, = the shard's row count on a sequence-parallel rank)
    const float2* tab_tok;      // optional: [f*h*w][64] — every token's 64 pairs gathered from the axis tables (16-byte aligned); the DiT builds it once per grid
};
// Sequence-parallel send layout (svi_hip/sequence_parallel.py): instead of in place, operand p of the q | k launch is stored as
// out[p][g][j][row][cg] — destination rank j = col / Dp owns head-channel block [j*Dp, (j+1)*Dp), inside it head group g = (col % Dp) / Dg,
// cg = col % Dg — so that every (operand, head group)'s all-to-all input is one contiguous [P][rows * Dg] tensor.
// rows_per_sample > 0 (the stacked CFG pair on a shard): the launch's rows are nb = rows / rows_per_sample samples (CFG branches) of that many rows one under
// the other, and a token's nb branches sit side by side: out[p][g][j][row % rows_per_sample][branch][cg].  What a destination receives from all sources is
// then token-major [L, nb * Dg] per (operand, head group): the two branches' heads are 2 x as many heads of ONE attention launch.
struct SviScatter { bf16* out0; bf16* out1; int P, Dp, Dg; int rows_per_sample; };
// out_scale multiplies the result before its single final rounding (the DiT folds the attention scale into q there)
svi_status svi_launch_rmsnorm_rope(bf16* x, int ld, int rows, int dim, const bf16* weight, float eps,
                                   const SviRope* rope, float out_scale, hipStream_t st);
svi_status svi_launch_rmsnorm_rope2(bf16* x, int ld, int rows, int dim, const bf16* weight, const bf16* weight1, float eps,
                                    const SviRope* rope, float out_scale, float out_scale1, hipStream_t st, const SviScatter* scatter = nullptr);
// The same launch writing MX e4m3 rows + block scales for both operands INSTEAD of the bf16 rows (operand 0 -> q8 / qs, operand 1 -> k8 / ks): the bits
// svi_launch_mx8_quantize makes of the bf16 rows.  Only the DiT's shapes (svi_rmsnorm_rope_q8_ok); rope is required.
bool svi_rmsnorm_rope_q8_ok(int dim, const SviRope* rope);
svi_status svi_launch_rmsnorm_rope2_q8(bf16* x, int ld, int rows, int dim, const bf16* weight, const bf16* weight1, float eps, const SviRope* rope, float out_scale,
                                       float out_scale1, hipStream_t st, const SviQk8& out);
// receive side of the two exchanges: V^T pieces [P(src)][Dp][lds] -> [Dp][L8] (token axis = source-major), attention output pieces
// [G][P(src = owner of head block)][Ls][Dg] -> [Ls][P*Dp] token rows
svi_status svi_launch_sp_unpack_vt(const bf16* recv, bf16* out, int P, int Dp, int Ls, int lds, int L8, hipStream_t st, int nb = 1, int Dg = 0);
svi_status svi_launch_sp_unpack_out(const bf16* recv, bf16* out, int P, int G, int Ls, int Dg, hipStream_t st, int nb = 1);
svi_status svi_launch_transpose(const bf16* in, int ldi, bf16* out, int ldo, int rows, int cols, hipStream_t st);
svi_status svi_launch_cfg_step(bf16* lat, const bf16* cond, const bf16* uncond, int64_t n, float s, float dsigma,
                               hipStream_t st);
svi_status svi_launch_add_bf16(bf16* a_inout, const bf16* b, int64_t n, hipStream_t st);
svi_status svi_launch_cfg3_step(bf16* lat, const bf16* cond, const bf16* uncond, const bf16* drop, int64_t n, float st, float sa, float dsigma,
                                hipStream_t stream);
svi_status svi_launch_video_to_u8(const float* video, unsigned char* out, long thw, hipStream_t st);
svi_status svi_launch_u8_to_video(const unsigned char* frames, float* out, int n, long hw, hipStream_t st);
svi_status svi_launch_sub_bf16(bf16* out, const bf16* a, const bf16* b, int64_t n, hipStream_t st);
svi_status svi_launch_fp8_e4m3_to_bf16(const unsigned char* in, bf16* out, int64_t n, hipStream_t st);
// fp32 C[M,N] = A[M,K] W[N,K]^T + bias[N] (+ res[M,N]) on the exact-fp32 MFMA kernel of the VAE (csrc/svi_vae.hip)
svi_status svi_launch_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K,
                               const float* res, int ldres, hipStream_t st);
