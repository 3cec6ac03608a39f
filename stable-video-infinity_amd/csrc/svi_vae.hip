// svi_vae.hip — the Wan 3-D causal VAE (encode / decode) on gfx950, fp32.
//
// Stands in for WanVideoVAE.encode/.decode -> VideoVAE_.encode/.decode
// (models/wan_video_vae.py:759-789, 525-575) and the blocks under them (:33-481).
//
// MI355X-first restructuring (results identical, see tests/test_gpu_vae.py):
//   * The reference streams the clip through the network in temporal chunks with a 2-frame feature cache per
//     causal conv because it targets 24-80 GB GPUs.  288 GB of HBM holds the WHOLE clip at every layer
//     (largest activation at 81f@832x480: [81,480,832,96] fp32 = 12.4 GB), so every CausalConv3d here is one
//     conv over the full sequence with (kt-1) zero frames in front — no cache, no torch.cat, no per-chunk
//     relaunch.  The two first-frame special cases of the chunked algorithm are kept explicitly:
//       - decoder upsample3d (:122-156): frame 0 skips time_conv and is hidden from later frames' history;
//       - encoder downsample3d (:162-173): frame 0 passes through; then a stride-2 un-padded time conv.
//   * Activations are channels-last [T, H, W, C] fp32: the contraction axis (input channels) is contiguous,
//     so every 3x3x3 conv is an implicit GEMM  out[pixel, co] = sum_tap sum_ci in[pixel+tap, ci] * w[tap, co, ci]
//     with 128-byte activation rows feeding the MFMA directly.
//   * The pipelines run the VAE in fp32 on purpose (pipelines/svi_video.py:386-387; docs/DevLog tip 4), so the
//     matrix core used is v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the 157 TFLOP/s fp32 matrix rate.
//
// Kernel: implicit-GEMM conv, tile = 128 output pixels x (32*NT) output channels per 256-thread workgroup
// (wave w owns pixels 32w..32w+31 x all NT column tiles), K loop over taps x 32-channel chunks, LDS double
// buffered through registers, XOR-swizzled 128-byte rows (conflict-free ds_read_b128).  One ds_read_b128
// per operand feeds 4 MFMA k-steps (lane-half hi owns k = 8q+4hi+s; A and B use the same map).
// Algorithmic work per conv: 2 * To*Ho*Wo * Cout * Cin * kt*kh*kw FLOP.
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svi_common.h"

namespace {

struct ConvP {
    const float* in; int Ti, Hi, Wi, Cin, ld_in;     // stored input dims; ld_in = floats between pixels
    const float* w; int ld_w;                        // packed [tap][Cout][ld_w] (ld_w >= Cin)
    const bf16* w3; int ld_w3; long plane_w3;        // the same weights as three bf16 planes (hi | mid | lo), [plane][tap][Cout][ld_w3]
    const void* w2h; const float* w2_inv;            // and as two fp16 planes (hi | lo) of w * 2^k(cout), same layout; w2_inv[cout] = 2^-k
    float in_scale;                                  // > 0: a power of two s with |in| * s <= 2^15 guaranteed by the producer (norm + SiLU): the fp16 two-term kernel may run
    const float* bias;
    float* out; int To, Ho, Wo, Cout, ld_out;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int ups;                                         // read input through a nearest x2 spatial upsample
    int zero_frame0;                                 // input frame 0 reads as zeros
    int t_begin;                                     // first output frame computed
    int out_mode;                                    // 0: out[t,y,x,co]   1: time interleave (see epilogue)
    int abl;                                         // timing ablations of conv_igemm_x3_kernel (SVI_VAE_ABL; results wrong): 1 no global loads, 2 no LDS stores, 4 no MFMAs, 8 no epilogue, 16 no K loop
    int t_out_off;                                   // added to the output frame index (mode 0)
    const float* res; int ld_res;                    // optional residual, same pixel indexing as out (mode 0)
    int act_silu;                                    // conv_igemm_kernel, mode 0: out = silu(conv + bias)  (pose embedder)
    bf16* out_bf16;                                  // conv_igemm_kernel, mode 0: store bf16 [pixel][ld_out] here instead of fp32 `out`
    const unsigned short* in_h; const unsigned short* in_l;      // conv_dma2h_kernel: the input as two fp16 planes [pixel][ld_in] of x * in_scale (hi | lo)
    int ord_T, ord_Lf, ord_G;                        // conv_dma2h_kernel: workgroup -> tile order (launch_conv_planes); ord_T = 0: tiles in pixel order
    int up_phase;                                    // conv_igemm_x3_kernel, mode 0: 1 + 2 py + px = this launch computes output pixels (2 y + py, 2 x + px) of an image twice the size
                                                     // of its (To, Ho, Wo) grid (one phase of a convolution behind a nearest x2 upsample, see launch_conv_up_phases); 0 = plain
};

__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvP p) {
    constexpr int A_BYTES = 128 * 128;               // 128 pixels x 32 floats
    constexpr int W_BYTES = NT * 32 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long HoWo = (long)p.Ho * p.Wo;
    const long P_total = (long)p.To * HoWo;
    const long p0 = (long)p.t_begin * HoWo + (long)blockIdx.x * 128;
    const int co0 = blockIdx.y * 32 * NT;
    const int Hv = p.ups ? 2 * p.Hi : p.Hi, Wv = p.ups ? 2 * p.Wi : p.Wi;

    // ---- staging assignment: activations 128 rows x 8 chunks (4 / thread); weights 32*NT rows x 8 chunks (NT / thread)
    const int ld_chunk = tid & 7;
    int a_t[4], a_y[4], a_x[4];
    bool a_ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long pp = p0 + (tid >> 3) + 32 * j;
        a_ok[j] = pp < P_total;
        const long q = a_ok[j] ? pp : 0;
        a_t[j] = (int)(q / HoWo);
        const int rem = (int)(q - (long)a_t[j] * HoWo);
        a_y[j] = rem / p.Wo;
        a_x[j] = rem - a_y[j] * p.Wo;
    }
    const int nchunk = (p.Cin + 31) >> 5;
    const int ntaps = p.kt * p.kh * p.kw;
    const int nk = ntaps * nchunk;
    f32x4 ra[4], rw[NT];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kidx) {
        const int tap = kidx / nchunk, cc = kidx - tap * nchunk;
        const int ta = tap / (p.kh * p.kw), tb = (tap / p.kw) % p.kh, tc = tap % p.kw;
        const int c = cc * 32 + ld_chunk * 4;
        const bool cin = c < p.Cin;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ti = a_t[j] * p.st + ta - p.pt;
            int yi = a_y[j] * p.sh + tb - p.ph, xi = a_x[j] * p.sw + tc - p.pw;
            bool ok = a_ok[j] && cin && ti >= 0 && ti < p.Ti && yi >= 0 && yi < Hv && xi >= 0 && xi < Wv;
            if (p.zero_frame0 && ti == 0) ok = false;
            if (p.ups) { yi >>= 1; xi >>= 1; }
            ra[j] = ok ? *reinterpret_cast<const f32x4*>(p.in + (((long)ti * p.Hi + yi) * p.Wi + xi) * p.ld_in + c) : zero4;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = co0 + (tid >> 3) + 32 * j;
            rw[j] = (cin && co < p.Cout) ? *reinterpret_cast<const f32x4*>(p.w + ((long)tap * p.Cout + co) * p.ld_w + c) : zero4;
        }
    };
    auto store_tile = [&](int buf) {
        char* As = smem + buf * (A_BYTES + W_BYTES);
        char* Ws = As + A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(As + tile_off((tid >> 3) + 32 * j, ld_chunk)) = ra[j];
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(Ws + tile_off((tid >> 3) + 32 * j, ld_chunk)) = rw[j];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        const int cur = k & 1;
        if (k + 1 < nk) load_tile(k + 1);
        const char* As = smem + cur * (A_BYTES + W_BYTES);
        const char* Ws = As + A_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(As + tile_off(32 * wave + l31, 2 * q + hi));
            f32x4 bv[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = *reinterpret_cast<const f32x4*>(Ws + tile_off(32 * n + l31, 2 * q + hi));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[n][s], acc[n], 0, 0, 0);
        }
        if (k + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds column co = co0 + 32n + l31, rows pixel = 32*wave + (r&3) + 8*(r>>2) + 4*hi
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = co0 + 32 * n + l31;
        if (co >= p.Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long pp = p0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (pp >= P_total) continue;
            float v = acc[n][r] + bv;
            if (p.out_mode == 0) {
                const long po = pp + (long)p.t_out_off * HoWo;
                if (p.res) v += p.res[po * p.ld_res + co];
                if (p.act_silu) v = v / (1.0f + expf(-v));
                if (p.out_bf16) p.out_bf16[po * p.ld_out + co] = (bf16)v;
                else p.out[po * p.ld_out + co] = v;
            } else {
                // upsample3d time_conv (vae:153-156): output channel halves become two consecutive frames
                const int half = p.Cout >> 1;
                const int t = (int)(pp / HoWo);
                const long sp = pp - (long)t * HoWo;
                const int j = co >= half ? 1 : 0;
                const long po = (long)(1 + 2 * (t - 1) + j) * HoWo + sp;
                p.out[po * p.ld_out + (co - j * half)] = v;
            }
        }
    }
}


// =================================================================================================
// fp32 convolution on the bf16 matrix cores: every fp32 operand is split EXACTLY-ish into three bf16 terms
//   x = hi + mid + lo   (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); 3 x 8 = 24 mantissa bits)
// and a product is evaluated as the six partial products of weight >= 2^-16 relative
//   a·w ~= ah·wh + ah·wm + am·wh + ah·wl + al·wh + am·wm     (dropped: 2^-24 relative and below)
// all accumulated in the MFMA's fp32 accumulator.  Six v_mfma_f32_32x32x16_bf16 (16 channels, 32 cycles each) replace eight
// v_mfma_f32_32x32x2_f32 (2 channels, 64 cycles each): 2.7x less matrix-pipe time at fp32-class accuracy (the parity tests
// hold the same 2e-5 / 2e-4 bounds as for the exact-fp32 kernel).  The reference runs the VAE in fp32 on purpose
// (pipelines/svi_video.py:386-387); this keeps that contract.
//
// Tile: 256 output pixels x 96 output channels per 512-thread workgroup (wave w owns pixels 32w..32w+31 x 3 channel tiles),
// K loop over taps x 32-channel chunks, LDS double buffer = 2 x (3 planes x [256][32] bf16 + 3 planes x [96][32] bf16) = 132 KiB.
// Activations are split on the fly while they are staged (registers -> LDS), weights were split once at bind time.
// The weight fragment is the MFMA A operand, the activation fragment the B operand, so a lane's accumulator holds 4
// consecutive output channels of ONE pixel: 16-byte channels-last stores.
// LDS rows are 64 B (32 bf16); 16-byte chunk c of row r sits at c ^ ((r >> 2) & 3) (conflict-free ds_read_b128).
// Where it stands: 1.65e15 bf16 MFMA FLOP per C2 decode in 1.6 s = 1.03 PFLOP/s — the same power-limited regime as the GEMMs
// (profiles/r1e_power_probe.txt).  Tried on top and dropped: splitting the layer input ONCE into three bf16 planes (a streaming
// pre-pass) and staging them by LDS-DMA with per-lane gather addresses and zero padding by the buffer range (no staging
// registers, no ds_write, 6 fewer VALU per MFMA): bit-identical, the convolutions ran at the same speed and the pre-pass (110 ms)
// and the 1.5x larger operand stream made the decode 5 % slower; six accumulator chains instead of three: no change.
// =================================================================================================
// ---- the same on fp16 words (template parameter H2) --------------------------------------------------------------------------
// fp16 carries 11 significant bits, so TWO words already hold 22 bits and three partial products
//   a·w ~= ah·wh + ah·wl + al·wh     (dropped: al·wl, 2^-22 relative)
// give the accuracy the six bf16 products give — at half the matrix-pipe work (v_mfma_f32_32x32x16_f16 runs at the bf16 rate).
// What fp16 lacks is range, so both operands are brought into it by exact power-of-two scales: weights per output channel at bind
// time (row maximum -> [2^13, 2^14)), activations by a scale the PRODUCER guarantees: RMS_norm bounds every output by
// sqrt(C)·max|gamma| and SiLU does not increase magnitudes, so the two 3x3x3 convolutions of every residual block, the head
// convolutions and to_qkv — 94 % of the decoder's FLOP — have a static bound (Tens::bound).  The scales are undone in the epilogue
// by one exact multiplication.  Convolutions whose input has no such bound (conv_in, shortcuts, resample / time convolutions,
// attention proj) stay on the three-term bf16 kernel.  SVI_VAE_X2H=0 sends everything to the three-term kernel (A/B).
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
__device__ __forceinline__ unsigned short bits16(bf16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ unsigned short bits16(f16 v) { return __builtin_bit_cast(unsigned short, v); }

#define X3_PIX 256
#define X3_CO 96
#define X3_A_PLANE (X3_PIX * 64)                 // 16 KiB
#define X3_W_PLANE (X3_CO * 64)                  // 6 KiB
#define X3_STAGE (3 * X3_A_PLANE + 3 * X3_W_PLANE)
#define X2H_STAGE (2 * X3_A_PLANE + 2 * X3_W_PLANE)
__device__ __forceinline__ int x3_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// split 4 fp32 into three bf16x4 terms
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16 hh = (bf16)x[e];
        const float r1 = x[e] - (float)hh;
        const bf16 mm = (bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (bf16)r2;
    }
}

// split 4 fp32 (already scaled into fp16 range) into two fp16x4 terms
__device__ __forceinline__ void split2h(const f32x4 x, u16x4& h, u16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f16 hh = (f16)x[e];
        h[e] = bits16(hh);
        l[e] = bits16((f16)(x[e] - (float)hh));
    }
}

// the same for one pair inside the pinned K step, 5 instructions: hi = f16(x s) by v_fma_mixlo/hi_f16 (multiply and conversion
// in one), lo = f16(x s - hi) by v_fma_mix_f32 with hi entering as an f16 operand, and one packed conversion
__device__ __forceinline__ void split2h_pair(float x0, float x1, float s, unsigned& h, unsigned& l) {
    float r0, r1;
    asm("v_fma_mixlo_f16 %[h], %[x0], %[s], 0\n\t"
        "v_fma_mixhi_f16 %[h], %[x1], %[s], 0\n\t"
        "v_fma_mix_f32 %[r0], %[x0], %[s], -%[h] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %[r1], %[x1], %[s], -%[h] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_cvt_pk_f16_f32 %[l], %[r0], %[r1]"
        : [h] "=&v"(h), [l] "=v"(l), [r0] "=&v"(r0), [r1] "=&v"(r1) : [x0] "v"(x0), [x1] "v"(x1), [s] "s"(s));
}

// timing ablations (tools/vae_ab.py, SVI_VAE_ABL): compiled in only with -DSVI_ABLATIONS so that the product kernel has no
// branches inside a K step
#ifdef SVI_ABLATIONS
#define SVI_X3_ABL(bit) (p.abl & (bit))
#else
#define SVI_X3_ABL(bit) false
#endif
template <bool H2>
__global__ __launch_bounds__(512, 2) void conv_igemm_x3_kernel(ConvP p) {
    constexpr int NPL = H2 ? 2 : 3;                                   // operand planes
    constexpr int STAGE = NPL * (X3_A_PLANE + X3_W_PLANE);
    constexpr int NWID = NPL * 384;                                   // weight vectors per stage: planes x 96 rows x 4 chunks
    constexpr int NWV = (NWID + 511) / 512;                           // ... per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long HoWo = (long)p.Ho * p.Wo;
    const long P_total = (long)p.To * HoWo;
    const long p0 = (long)p.t_begin * HoWo + (long)blockIdx.x * X3_PIX;
    const int co0 = blockIdx.y * X3_CO;
    const int Hv = p.ups ? 2 * p.Hi : p.Hi, Wv = p.ups ? 2 * p.Wi : p.Wi;
    const int ups_sh = p.ups ? 1 : 0;
    const int nchunk = (p.Cin + 31) >> 5;
    const int khw = p.kh * p.kw, ntaps = p.kt * khw;
    const int nk = ntaps * nchunk;

    // ---- staging assignment: activations 256 rows x 8 float4 (4 per thread); weights 3 planes x 96 rows x 4 chunks (16 B).
    // Everything that depends on the pixel only is worked out ONCE: its input coordinates for tap (0,0,0) and a bit mask of the
    // taps whose input position exists (inside the tensor, not the zeroed frame).  Per K step that leaves an add, a multiply-
    // add and a select per pixel; positions that do not exist are read at an offset beyond the buffer's range, which the
    // buffer load returns as zeros — no branches, so hipcc can count the loads in flight and the staged tile can be prefetched
    // a whole K step ahead (two register sets), instead of being waited for right behind the MFMAs that were meant to cover it.
    const int a_c4 = tid & 7;                        // which float4 (4 channels) of the 32-channel chunk
    const int t_first = (int)(min(p0, P_total - 1) / HoWo);      // pixels are t-major: the tile's first pixel has the smallest t
    const int t_base = max(t_first * p.st - p.pt, 0);             // buffer base = that input frame (32-bit offsets from there)
    int bt[4], by[4], bx[4];
    unsigned a_mask[4], base_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long pp = p0 + (tid >> 3) + 64 * j;
        const bool pix = pp < P_total;
        const long q = pix ? pp : 0;
        const int at = (int)(q / HoWo);
        const int rem = (int)(q - (long)at * HoWo);
        const int ay = rem / p.Wo, ax = rem - ay * p.Wo;
        bt[j] = at * p.st - p.pt; by[j] = ay * p.sh - p.ph; bx[j] = ax * p.sw - p.pw;
        unsigned m = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            const int ta = tap / khw, tb = (tap / p.kw) % p.kh, tc = tap % p.kw;
            const int ti = bt[j] + ta, yi = by[j] + tb, xi = bx[j] + tc;
            bool ok = pix && ti >= 0 && ti < p.Ti && yi >= 0 && yi < Hv && xi >= 0 && xi < Wv;
            if (p.zero_frame0 && ti == 0) ok = false;
            if (ok) m |= 1u << tap;
        }
        a_mask[j] = m;
        bt[j] -= t_base;
        base_off[j] = (unsigned)(((bt[j] * p.Hi + by[j]) * p.Wi + bx[j]) * p.ld_in) * 4u;      // tap (0,0,0); only used by the two-term form
    }
    const long base_el = (long)t_base * p.Hi * p.Wi * p.ld_in;
    const long rem_bytes = ((long)p.Ti * p.Hi * p.Wi * p.ld_in - base_el) * 4;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + base_el), 0,
                                                                           (int)(unsigned)min(rem_bytes, H2 ? 0xFFE00000L : 0xFFFFF000L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(H2 ? const_cast<void*>(p.w2h) : (void*)const_cast<bf16*>(p.w3), 0,
                                                                          (int)(unsigned)min((long)NPL * p.plane_w3 * 2, H2 ? 0xFFE00000L : 0xFFFFF000L), 0x00020000);
    const float a_scale = H2 ? p.in_scale : 1.0f;
    // an offset no buffer window reaches.  Two-term form: low enough that adding a channel offset (< 64 KiB) cannot wrap, so a tap
    // that does not exist needs no select per K step (launch_conv keeps the windows below it)
    const unsigned OOB = H2 ? 0xFFF00000u : 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);   // zero records: every load reads zeros
    unsigned woff[NWV];                              // byte offset of this thread's weight chunks at tap 0, channel chunk 0
    int wch[NWV];
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
        const int id = tid + 512 * i;                // (plane, row, chunk): NPL x 96 x 4
        const int pl = id / 384, rem = id - pl * 384, row = rem >> 2, ch = rem & 3;
        const bool ok = id < NWID && co0 + row < p.Cout;
        woff[i] = ok ? (unsigned)((pl * p.plane_w3 + (long)(co0 + row) * p.ld_w3 + ch * 8) * 2) : OOB;
        wch[i] = ch * 8;
    }
    f32x4 ra[2][4];
    u32x4 rw[2][NWV];                                // two register sets: the tile being staged and the one in flight

    // One K step, written as six slices so that the work of three K steps overlaps inside one basic block: slice s issues the
    // 6 MFMAs of (k-step ks = s / 3, cout block n = s % 3) of the step held in LDS stage CBUF, then splits / stores ONE staged
    // vector of the step in register set SS (one step ahead) into LDS stage SBUF, then requests ONE vector of step `kidx` (two
    // steps ahead) into register set SI.  A step past the end (valid = false) loads out-of-range zeros, stages zeros, adds zeros.
#define SVI_X3_LOAD_A(S, j)                                                                                                      \
    do {                                                                                                                         \
        /* tap_off[j]: byte offset of this pixel's input position for the tap being requested (OOB if it does not exist),      \
           recomputed only when the tap changes (SVI_X3_ADVANCE); per step just the channel offset is added */                 \
        const unsigned off_ = tap_off[j] + (unsigned)c_ * 4u;                                                                    \
        if constexpr (H2) {   /* Cin % 32 == 0: c_ < Cin in every real step; a step past the end reads through rs_none */         \
            ra[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wvalid_ ? rs_in : rs_none, off_, 0, 0));  \
        } else {                                                                                                                 \
            const bool ok_ = cin_ & (tap_off[j] != OOB);                                                                         \
            ra[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok_ ? off_ : OOB, 0, 0));          \
        }                                                                                                                        \
    } while (0)
#define SVI_X3_LOAD_W(S, i)                                                                                                      \
    do {                                                                                                                         \
        if constexpr ((i) < NWV) {                                                                                               \
            const bool ok_ = wvalid_ && woff[i] != OOB && cc_ * 32 + wch[i] < p.ld_w3;                                           \
            rw[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, ok_ ? woff[i] + wk_ : OOB, 0, 0);                             \
        }                                                                                                                        \
    } while (0)
#define SVI_X3_STAGE_A(S, buf, j)                                                                                                \
    do {                                                                                                                         \
        char* As_ = smem + (buf) * STAGE;                                                                                        \
        const int off_ = x3_off((tid >> 3) + 64 * (j), a_c4 >> 1) + (a_c4 & 1) * 8;                                              \
        if constexpr (H2) {                                                                                                      \
            u16x4 h_, l_;                                                                                                        \
            split2h(ra[S][j] * a_scale, h_, l_);                                                                                 \
            *reinterpret_cast<u16x4*>(As_ + off_) = h_;                                                                          \
            *reinterpret_cast<u16x4*>(As_ + X3_A_PLANE + off_) = l_;                                                             \
        } else {                                                                                                                 \
            bf16x4 h_, m_, l_;                                                                                                   \
            split3(ra[S][j], h_, m_, l_);                                                                                        \
            *reinterpret_cast<bf16x4*>(As_ + off_) = h_;                                                                         \
            *reinterpret_cast<bf16x4*>(As_ + X3_A_PLANE + off_) = m_;                                                            \
            *reinterpret_cast<bf16x4*>(As_ + 2 * X3_A_PLANE + off_) = l_;                                                        \
        }                                                                                                                        \
    } while (0)
#define SVI_X3_STAGE_W(S, buf, i)                                                                                                \
    do {                                                                                                                         \
        const int id_ = tid + 512 * (i);                                                                                         \
        if ((i) < NWV && id_ < NWID) {                                                                                           \
            const int pl_ = id_ / 384, rem_ = id_ - pl_ * 384, row_ = rem_ >> 2, ch_ = rem_ & 3;                                 \
            *reinterpret_cast<u32x4*>(smem + (buf) * STAGE + NPL * X3_A_PLANE + pl_ * X3_W_PLANE + x3_off(row_, ch_)) = rw[S][(i) < NWV ? (i) : 0]; \
        }                                                                                                                        \
    } while (0)
#define SVI_X3_TAP_BASES()                                                                                                       \
    if constexpr (H2) {   /* no upsampled read here (launch_conv): a tap moves every pixel by the same (scalar) byte distance */ \
        const unsigned d_ = (unsigned)(((it_ta * p.Hi + it_tb) * p.Wi + it_tc) * p.ld_in) * 4u;                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) tap_off[j] = ((a_mask[j] >> it_tap) & 1u) ? base_off[j] + d_ : OOB;        \
    } else {                                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                          \
            const int yi_ = (by[j] + it_tb) >> ups_sh, xi_ = (bx[j] + it_tc) >> ups_sh;                                          \
            const unsigned o_ = (unsigned)((((bt[j] + it_ta) * p.Hi + yi_) * p.Wi + xi_) * p.ld_in) * 4u;                        \
            tap_off[j] = ((a_mask[j] >> it_tap) & 1u) ? o_ : OOB;                                                                \
        }                                                                                                                        \
    }
#define SVI_X3_ADVANCE()                                                                                                         \
    do {                                                                                                                         \
        it_wk += 64u;                                                                                                            \
        if (++it_cc == nchunk) {                                                                                                 \
            it_cc = 0; ++it_tap;                                                                                                 \
            it_wk = (unsigned)((long)it_tap * p.Cout * p.ld_w3 * 2);                                                             \
            if (++it_tc == p.kw) { it_tc = 0; if (++it_tb == p.kh) { it_tb = 0; ++it_ta; } }                                     \
            SVI_X3_TAP_BASES();                                                                                                  \
        }                                                                                                                        \
    } while (0)

    // One K step with every instruction's place fixed (sched_barrier between the pieces): each MFMA is followed by a piece of the
    // staging / request work — elements of the operand split, the LDS stores, the address arithmetic and the buffer load — so that
    // the vector ALU works in the MFMAs' shadows and the MFMAs chained on one accumulator are spaced apart.  The fragments of slice
    // s+1 are read behind the first MFMA of slice s.  (Letting hipcc order a slice itself: 1.79 vs 1.73 s per C2 decode, r1.)
    // Three-term bf16 form: six MFMAs per slice; two-term fp16 form (H2): three.
#define SVI_SB() __builtin_amdgcn_sched_barrier(0)
#define SVI_X3_SPLIT1(S, j, e)                                                                                                   \
    do {                                                                                                                         \
        if constexpr (H2) {                                                                                                      \
            /* hi = f16(x s), lo = f16(x s - hi): one v_fma_mixlo/hi_f16 and one v_fma_mix_f32 (hi enters as an f16 operand) */  \
            const float x_ = ra[S][j][e];                                                                                        \
            const f16 hh_ = (f16)(x_ * a_scale);                                                                                 \
            hq_[e] = bits16(hh_); mq_[e] = bits16((f16)__builtin_fmaf(x_, a_scale, -(float)hh_));                                \
        } else {                                                                                                                 \
            const float x_ = ra[S][j][e];                                                                                        \
            const bf16 hh_ = (bf16)x_;                                                                                           \
            const float r1_ = x_ - (float)hh_;                                                                                   \
            const bf16 mm_ = (bf16)r1_;                                                                                          \
            const float r2_ = r1_ - (float)mm_;                                                                                  \
            hq_[e] = bits16(hh_); mq_[e] = bits16(mm_); lq_[e] = bits16((bf16)r2_);                                              \
        }                                                                                                                        \
    } while (0)
#define SVI_X3_MFMA(WP, AP)                                                                                                      \
    do {                                                                                                                         \
        if constexpr (H2) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf_[wc][WP]), __builtin_bit_cast(f16x8, af_[ks][AP]), acc[n], 0, 0, 0);\
        else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_[wc][WP], af_[ks][AP], acc[n], 0, 0, 0);                        \
    } while (0)
#define SVI_X3_STORES(SS, SBUF)                                                                                                          \
    do {                                                                                                                         \
        if (!SVI_X3_ABL(2)) {                                                                                                    \
            if (s_ < 4) {                                                                                                        \
                char* As_ = smem + (SBUF) * STAGE;                                                                               \
                const int off_ = x3_off((tid >> 3) + 64 * s_, a_c4 >> 1) + (a_c4 & 1) * 8;                                       \
                if constexpr (H2) {                                                                                              \
                    *reinterpret_cast<u32x2*>(As_ + off_) = u32x2{hp_[0], hp_[1]};                                               \
                    *reinterpret_cast<u32x2*>(As_ + X3_A_PLANE + off_) = u32x2{lp_[0], lp_[1]};                                  \
                } else {                                                                                                         \
                    *reinterpret_cast<u16x4*>(As_ + off_) = hq_;                                                                 \
                    *reinterpret_cast<u16x4*>(As_ + X3_A_PLANE + off_) = mq_;                                                    \
                    *reinterpret_cast<u16x4*>(As_ + 2 * X3_A_PLANE + off_) = lq_;                                                \
                }                                                                                                                \
            } else if (s_ == 4) { SVI_X3_STAGE_W(SS, SBUF, 0); SVI_X3_STAGE_W(SS, SBUF, 1); }                                    \
            else SVI_X3_STAGE_W(SS, SBUF, 2);                                                                                    \
        }                                                                                                                        \
    } while (0)
#define SVI_X3_LOADS(SI)                                                                                                           \
    do {                                                                                                                         \
        if (s_ == 0) SVI_X3_LOAD_A(SI, 0);                                                                                       \
        else if (s_ == 1) SVI_X3_LOAD_A(SI, 1);                                                                                  \
        else if (s_ == 2) SVI_X3_LOAD_A(SI, 2);                                                                                  \
        else if (s_ == 3) SVI_X3_LOAD_A(SI, 3);                                                                                  \
        else if (s_ == 4) { SVI_X3_LOAD_W(SI, 0); SVI_X3_LOAD_W(SI, 1); }                                                        \
        else SVI_X3_LOAD_W(SI, 2);                                                                                               \
    } while (0)
#define SVI_X3_STEP(CBUF, SI, kidx, valid, SS, SBUF)                                                                             \
    do {                                                                                                                         \
        const int cc_ = it_cc;                                                                                                   \
        const int c_ = cc_ * 32 + a_c4 * 4;                                                                                      \
        const bool wvalid_ = (valid) && !SVI_X3_ABL(1);                                                                          \
        const bool cin_ = wvalid_ && c_ < p.Cin;                                                                                 \
        const unsigned wk_ = it_wk;                                                                                              \
        const char* As_c = smem + (CBUF) * STAGE;                                                                                \
        const char* Ws_c = As_c + NPL * X3_A_PLANE;                                                                              \
        bf16x8 af_[2][NPL], wf_[2][NPL];                                                                                         \
        _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                                                                     \
            af_[0][pl] = *reinterpret_cast<const bf16x8*>(As_c + pl * X3_A_PLANE + x3_off(32 * wave + l31, hi));                 \
            wf_[0][pl] = *reinterpret_cast<const bf16x8*>(Ws_c + pl * X3_W_PLANE + x3_off(l31, hi));                             \
        }                                                                                                                        \
        SVI_SB();                                                                                                                \
        _Pragma("unroll") for (int s_ = 0; s_ < 6; ++s_) {                                                                       \
            const int ks = s_ / 3, n = s_ % 3, sn = s_ + 1, ksn = sn / 3, nn = sn % 3;                                           \
            const int wc = s_ & 1, wn = wc ^ 1;                                                                                  \
            u16x4 hq_, mq_, lq_;                                                                                                 \
            unsigned hp_[2], lp_[2];                                                                                             \
            if constexpr (H2) SVI_X3_MFMA(1, 0);   /* wl ah */                                                                   \
            else SVI_X3_MFMA(1, 1);                /* wm am */                                                                   \
            SVI_SB();                                                                                                            \
            if (s_ < 5) {                                                                                                        \
                _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                                               \
                    wf_[wn][pl] = *reinterpret_cast<const bf16x8*>(Ws_c + pl * X3_W_PLANE + x3_off(32 * nn + l31, 2 * ksn + hi));\
                if (s_ == 1) {                                                                                                   \
                    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                                           \
                        af_[1][pl] = *reinterpret_cast<const bf16x8*>(As_c + pl * X3_A_PLANE + x3_off(32 * wave + l31, 2 + hi)); \
                }                                                                                                                \
            }                                                                                                                    \
            if constexpr (H2) { if (s_ < 4) split2h_pair(ra[SS][s_ < 4 ? s_ : 0][0], ra[SS][s_ < 4 ? s_ : 0][1], a_scale, hp_[0], lp_[0]); }\
            else { if (s_ < 4) SVI_X3_SPLIT1(SS, s_, 0); }                                                                       \
            SVI_SB();                                                                                                            \
            if constexpr (H2) {                                                                                                  \
                SVI_X3_MFMA(0, 1);                 /* wh al */                                                                   \
                SVI_SB();                                                                                                        \
                if (s_ < 4) split2h_pair(ra[SS][s_ < 4 ? s_ : 0][2], ra[SS][s_ < 4 ? s_ : 0][3], a_scale, hp_[1], lp_[1]);       \
                SVI_X3_STORES(SS, SBUF);                                                                                                 \
                SVI_SB();                                                                                                        \
            } else {                                                                                                             \
                SVI_X3_MFMA(2, 0);                 /* wl ah */                                                                   \
                SVI_SB();                                                                                                        \
                if (s_ < 4) SVI_X3_SPLIT1(SS, s_, 1);                                                                            \
                SVI_SB();                                                                                                        \
                SVI_X3_MFMA(0, 2);                 /* wh al */                                                                   \
                SVI_SB();                                                                                                        \
                if (s_ < 4) SVI_X3_SPLIT1(SS, s_, 2);                                                                            \
                SVI_SB();                                                                                                        \
                SVI_X3_MFMA(1, 0);                 /* wm ah */                                                                   \
                SVI_SB();                                                                                                        \
                if (s_ < 4) SVI_X3_SPLIT1(SS, s_, 3);                                                                            \
                SVI_SB();                                                                                                        \
                SVI_X3_MFMA(0, 1);                 /* wh am */                                                                   \
                SVI_SB();                                                                                                        \
                SVI_X3_STORES(SS, SBUF);                                                                                                 \
                SVI_SB();                                                                                                        \
            }                                                                                                                    \
            SVI_X3_MFMA(0, 0);                     /* wh ah */                                                                   \
            SVI_SB();                                                                                                            \
            SVI_X3_LOADS(SI);                                                                                                      \
            SVI_SB();                                                                                                            \
        }                                                                                                                        \
        SVI_X3_ADVANCE();                                                                                                        \
    } while (0)

    f32x16 acc[3];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // prologue: step 0 -> set 0 -> stage 0; step 1 -> set 1.  Steady state: pairs of steps, no branch inside a step.
    int it_tap = 0, it_cc = 0, it_ta = 0, it_tb = 0, it_tc = 0;
    unsigned it_wk = 0;
    unsigned tap_off[4];
    SVI_X3_TAP_BASES();
    {
        const int cc_ = 0;
        const int c_ = a_c4 * 4;
        const bool wvalid_ = !SVI_X3_ABL(1), cin_ = wvalid_ && c_ < p.Cin;
        const unsigned wk_ = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) SVI_X3_LOAD_A(0, j);
        SVI_X3_LOAD_W(0, 0); SVI_X3_LOAD_W(0, 1); SVI_X3_LOAD_W(0, 2);
    }
    SVI_X3_ADVANCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) SVI_X3_STAGE_A(0, 0, j);
    SVI_X3_STAGE_W(0, 0, 0); SVI_X3_STAGE_W(0, 0, 1); SVI_X3_STAGE_W(0, 0, 2);
    {
        const int cc_ = it_cc;
        const int c_ = cc_ * 32 + a_c4 * 4;
        const bool wvalid_ = nk > 1 && !SVI_X3_ABL(1), cin_ = wvalid_ && c_ < p.Cin;
        const unsigned wk_ = it_wk;
#pragma unroll
        for (int j = 0; j < 4; ++j) SVI_X3_LOAD_A(1, j);
        SVI_X3_LOAD_W(1, 0); SVI_X3_LOAD_W(1, 1); SVI_X3_LOAD_W(1, 2);
    }
    SVI_X3_ADVANCE();
    __syncthreads();
    const int nkk = SVI_X3_ABL(16) ? 0 : nk;
    for (int k = 0; k < nkk; k += 2) {
        SVI_X3_STEP(0, 0, k + 2, k + 2 < nk, 1, 1);      // compute stage 0 | stage set 1 (step k+1) -> stage 1 | load step k+2 -> set 0
        __syncthreads();
        SVI_X3_STEP(1, 1, k + 3, k + 3 < nk, 0, 0);      // compute stage 1 | stage set 0 (step k+2) -> stage 0 | load step k+3 -> set 1
        __syncthreads();
    }
#undef SVI_X3_STEP
#undef SVI_X3_ADVANCE
#undef SVI_X3_TAP_BASES
#undef SVI_X3_LOAD_A
#undef SVI_X3_LOAD_W
#undef SVI_X3_STAGE_A
#undef SVI_X3_STAGE_W
#undef SVI_X3_SPLIT1
#undef SVI_X3_MFMA
#undef SVI_X3_STORES
#undef SVI_X3_LOADS

    // ---- epilogue: lane holds pixel pp = p0 + 32 wave + l31, channels co0 + 32 n + 8 rg + 4 hi + 0..3 in acc[n][4 rg + e].
    // The 12 bias vectors and 12 residual vectors of the lane are requested in one batch (one memory round trip instead of 24
    // dependent ones — the same fix as the GEMM epilogue's), then added and stored.
    const long pp = p0 + 32 * wave + l31;
    if (pp >= P_total) return;
    if (SVI_X3_ABL(8) && acc[0][0] != 123.456f) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bv[12], rv[12], sv[12];
    const bool with_res = p.out_mode == 0 && p.res;
    const float inv_a = H2 ? 1.0f / p.in_scale : 1.0f;                       // exact: powers of two
    long po = pp + (long)p.t_out_off * HoWo;
    if (!H2 && p.up_phase) {             // one phase of an upsample convolution: pixel (t, y, x) of this grid is (t, 2 y + py, 2 x + px) of the output
        const int ph_ = p.up_phase - 1;
        const int t = (int)(pp / HoWo);
        const int rem = (int)(pp - (long)t * HoWo);
        const int y = rem / p.Wo, x = rem - y * p.Wo;
        po = ((long)(t + p.t_out_off) * (2 * p.Ho) + 2 * y + (ph_ >> 1)) * (2 * p.Wo) + 2 * x + (ph_ & 1);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int co = co0 + 32 * (i >> 2) + 8 * (i & 3) + 4 * hi;
        const bool ok = co < p.Cout;
        bv[i] = (ok && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + co) : zero4;
        rv[i] = (ok && with_res) ? *reinterpret_cast<const f32x4*>(p.res + po * p.ld_res + co) : zero4;
        if constexpr (H2) sv[i] = ok ? *reinterpret_cast<const f32x4*>(p.w2_inv + co) * inv_a : zero4;
    }
    long pq0 = 0;
    const int half = p.Cout >> 1;
    if (p.out_mode != 0) {               // upsample3d time_conv (vae:153-156): output channel halves become two consecutive frames
        const int t = (int)(pp / HoWo);
        const long sp = pp - (long)t * HoWo;
        pq0 = (long)(1 + 2 * (t - 1)) * HoWo + sp;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int n = i >> 2, rg = i & 3;
        const int co = co0 + 32 * n + 8 * rg + 4 * hi;
        if (co >= p.Cout) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = acc[n][4 * rg + e];
            if constexpr (H2) a *= sv[i][e];
            v[e] = (a + bv[i][e]) + rv[i][e];
        }
        if (p.out_mode == 0) {
            *reinterpret_cast<f32x4*>(p.out + po * p.ld_out + co) = v;
        } else {
            const int j = co >= half ? 1 : 0;
            *reinterpret_cast<f32x4*>(p.out + (pq0 + (long)j * HoWo) * p.ld_out + (co - j * half)) = v;
        }
    }
}


// =================================================================================================
// Two-term fp16 convolution on the producer's fp16 word pairs, staged by LDS-DMA, with the three x-taps of a kernel row served from
// ONE staged strip (round 3).
//
// What bounds conv_igemm_x3_kernel<true> is neither the matrix pipe nor LDS: every (tap, 32-channel) K step re-fetches the tile's
// 256 pixel rows, so a 3x3x3 convolution pulls each input element 27 times through the CU's vector-memory path — 44 KiB per step and
// CU every ~1.6 us = ~7 TB/s chip-wide, which is the L2 -> CU fill rate this part sustains (MI355X_MICROARCH.md, ldsdma-fill: 6.4-6.8 TB/s).
// Removing the staging arithmetic alone (the producer writes hi = f16(x s), lo = f16(x s - hi) once, this kernel only moves bytes) bought
// 4 %; a three-stage DMA ring nothing.  So the bytes had to go:
//   * a K step is (frame tap, row tap, 32 channels); its activation strip is the tile's 256 consecutive pixels PLUS one pixel on either
//     side (272 LDS rows of 64 B per plane), fetched once and read three times: x-tap tc of output pixel i is LDS row i + tc.  Where
//     pixel i + tc - 1 is not the x-neighbour of pixel i (image border: the strip continues into the next image row) the lane's
//     fragment is replaced by zeros — the padding the convolution asks for;
//   * the weights of the three x-taps ride along (3 x 12 KiB): 70 DMA instructions per step and workgroup for 54 MFMAs per wave,
//     against 3 x 44 = 132 instructions for the same MFMAs before: 1.9 x fewer bytes through the fill path;
//   * a tap whose (frame, row) does not exist for a pixel is an offset beyond the descriptor's range: the DMA writes zeros.
// Summation order differs from the x3 kernels (x-taps innermost), so results agree to rounding, not bit for bit; the parity bounds are
// the VAE's (rel-L2 2e-5 / max-abs 2e-4 against the reference).  Layers: 3x3 spatial taps, stride 1, 'same' padding — every
// residual-block convolution.  LDS: two stages x (2 x [272][64 B] + 3 taps x 2 x [96][64 B]) = 140 KiB.
// =================================================================================================
#define D2_ROWS 272
#define D2_A_PLANE (D2_ROWS * 64)
#define D2_STAGE (2 * D2_A_PLANE + 6 * X3_W_PLANE)
// NB: 32-output-channel blocks per workgroup.  3: the residual-block convolutions (Cout = 96 per grid.y).  1: narrow heads (Cout <= 32: the
// decoder's 96 -> 3 and the encoder's 384 -> 32 head convolutions) — a third of the weight pieces and MFMAs, output channel count not
// necessarily a multiple of 4 (the lane that holds the last, partial group of four reads bias / scale element-wise and writes the pad
// channels of the [pixel][ld_out] row as zeros).
template <int NB>
__global__ __launch_bounds__(512, 2) void conv_dma2h_kernel(ConvP p) {
#ifdef SVI_D2_DENSE      // timing experiment (results wrong): every DMA piece reads 1 KiB of CONTIGUOUS memory, as a chunk-major plane layout would give
    p.ld_in = 32; p.ld_w3 = 32;
#endif
    constexpr int WP = 2 * NB;                       // 1 KiB weight pieces per (x-tap, plane): 32 NB rows of 64 B
    constexpr int NPIECE = 34 + 6 * WP;              // per K step: 34 activation pieces + the weights of three x-taps x two planes
    constexpr int D2_SLOTS = (NPIECE + 7) / 8;       // DMA instructions per wave and step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const long HoWo = (long)p.Ho * p.Wo;
    const long P_total = (long)p.To * HoWo;
    // Tile order.  Tiles are 256 consecutive pixels, frame-major; in pixel order the chip sweeps one frame before the next, so the two
    // earlier frames a causal 3x3x3 convolution reads were last touched one and two whole frames ago (153 MB of planes per frame at
    // 480 x 832 x 96: out of L2, largely out of the 256 MB MALL).  With ord_T frames of ord_Lf tiles each, workgroups walk groups of ord_G tiles
    // (about 16 image rows) through ALL frames before moving down the image: the slab of input rows a group needs from frames t-2, t-1, t is
    // reused while it is still close.
    // Output-channel blocks of one pixel tile sit next to each other in the launch order (1-D grid: workgroup = tile * blocks + block), so the
    // 2 or 4 workgroups that read the same activation strips run together instead of a whole tensor apart.
    const int ncob = (p.Cout + NB * 32 - 1) / (NB * 32);
    long tile = blockIdx.x / ncob;
    const int cob = (int)(blockIdx.x - tile * ncob);
    if (p.ord_T > 0) {
        const long per_group = (long)p.ord_T * p.ord_G;
        const int full = p.ord_Lf / p.ord_G;                     // whole groups; the rest of a frame forms a last, shorter group
        const long g = tile / per_group;
        int gl = p.ord_G;
        long r = tile - g * per_group, gbase = g * p.ord_G;
        if (g >= full) { gl = p.ord_Lf - full * p.ord_G; r = tile - (long)full * per_group; gbase = (long)full * p.ord_G; }
        const int t = (int)(r / gl), j = (int)(r - (long)t * gl);
        tile = (long)t * p.ord_Lf + gbase + j;
    }
    const long p0 = (long)p.t_begin * HoWo + tile * X3_PIX;
    const int co0 = cob * NB * 32;
    const int nchunk = p.Cin >> 5;
    const int nrow = p.kt * p.kh;                    // (frame tap, row tap) pairs
    const int nk = nrow * nchunk;

    const int t_first = (int)(max(min(p0 - 1, P_total - 1), 0L) / HoWo);
    const int t_base = max(t_first * p.st - p.pt, 0);
    // ---- per-slot DMA bookkeeping: slot i of wave w is piece q = w + 8 i (q < 34: activation strip, else weights)
    unsigned s_off[D2_SLOTS], s_mask[D2_SLOTS];
#pragma unroll
    for (int i = 0; i < D2_SLOTS; ++i) {
        const int q = wave + 8 * i;
        s_off[i] = 0xFFF00000u; s_mask[i] = 0;
        if (q < 34) {
            const int r16 = q % 17, row = r16 * 16 + (lane >> 2);
            const long pp = p0 - 1 + row;
            const bool pix = pp >= 0 && pp < P_total;
            const long qq = pix ? pp : 0;
            const int at = (int)(qq / HoWo);
            const int rem = (int)(qq - (long)at * HoWo);
            const int ay = rem / p.Wo, ax = rem - ay * p.Wo;
            const int bt = at * p.st - p.pt, by = ay - p.ph;
            unsigned m = 0;
            for (int rt = 0; rt < nrow; ++rt) {
                const int ta = rt / p.kh, tb = rt - ta * p.kh;
                const int ti = bt + ta, yi = by + tb;
                bool ok = pix && ti >= 0 && ti < p.Ti && yi >= 0 && yi < p.Hi;
                if (p.zero_frame0 && ti == 0) ok = false;
                if (ok) m |= 1u << rt;
            }
            s_mask[i] = m;
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            s_off[i] = (unsigned)((((bt - t_base) * p.Hi + by) * p.Wi + ax) * p.ld_in) * 2u + (unsigned)chunk * 16u;
        } else if (q < NPIECE) {
            const int w = q - 34, tc = w / (2 * WP), pl = (w % (2 * WP)) / WP, row = (w % WP) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            if (co0 + row < p.Cout) s_off[i] = (unsigned)((pl * p.plane_w3 + ((long)tc * p.Cout + co0 + row) * p.ld_w3 + chunk * 8) * 2);
        }
    }
    const long base_el = (long)t_base * p.Hi * p.Wi * p.ld_in;
    const long rem_bytes = ((long)p.Ti * p.Hi * p.Wi * p.ld_in - base_el) * 2;
    const int win = (int)(unsigned)min(rem_bytes, 0xFFE00000L);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.in_h + base_el), 0, win, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.in_l + base_el), 0, win, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2h), 0, (int)(unsigned)min((long)2 * p.plane_w3 * 2, 0xFFE00000L), 0x00020000);
    const unsigned OOB = 0xFFF00000u;
    // this lane's output pixel and whether its left / right x-neighbour exists
    bool left_ok, right_ok;
    {
        const long pp = min(p0 + 32 * wave + l31, P_total - 1);
        const int ax = (int)(pp % p.Wo);
        left_ok = ax > 0; right_ok = ax < p.Wo - 1;
    }

    int it_rt = 0, it_cc = 0, it_ta = 0, it_tb = 0;
    auto request = [&](int buf) {                    // K step (it_rt, it_cc) -> stage buf
        char* As = smem + buf * D2_STAGE;
        char* Ws = As + 2 * D2_A_PLANE;
        const unsigned d_act = (unsigned)(((it_ta * p.Hi + it_tb) * p.Wi) * p.ld_in) * 2u + (unsigned)it_cc * 64u;
        const unsigned d_w = (unsigned)((long)it_rt * p.kw * p.Cout * p.ld_w3 * 2) + (unsigned)it_cc * 64u;
#pragma unroll
        for (int i = 0; i < D2_SLOTS; ++i) {
            const int q = wave + 8 * i;
            if (q < 34) {
#ifdef SVI_D2_SKIP_A
                if (it_rt | it_cc) continue;
#endif
                const int pl = q / 17, r16 = q % 17;
                const unsigned off = ((s_mask[i] >> it_rt) & 1u) ? s_off[i] + d_act : OOB;
                if (pl == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (lptr_t)(As + r16 * 1024), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, (lptr_t)(As + D2_A_PLANE + r16 * 1024), 16, off, 0, 0, 0);
            } else if (q < NPIECE) {
#ifdef SVI_D2_SKIP_W
                if (it_rt | it_cc) continue;
#endif
                const int w = q - 34, tc = w / (2 * WP), pl = (w % (2 * WP)) / WP, r16 = w % WP;
                const unsigned off = s_off[i] == OOB ? OOB : s_off[i] + d_w;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(Ws + (tc * 2 + pl) * X3_W_PLANE + r16 * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto advance = [&]() {
        if (++it_cc == nchunk) {
            it_cc = 0; ++it_rt;
            if (++it_tb == p.kh) { it_tb = 0; ++it_ta; }
        }
    };

    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    request(0);
    advance();
    __syncthreads();                     // (hipcc drains the LDS-DMA with vmcnt(0) in front of the barrier)
    const f16x8 zero8 = {(f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0};
    for (int k = 0; k < nk; ++k) {
        const int cur = k & 1;
        if (k + 1 < nk) request(cur ^ 1);            // stage cur ^ 1 was last read in step k - 1: every wave has passed the barrier behind it
        advance();
        const char* As = smem + cur * D2_STAGE;
        const char* Ws = As + 2 * D2_A_PLANE;
        // six sub-steps (x-tap tc, k-half ks); the eight fragment reads of sub-step s + 1 are requested before the nine MFMAs of sub-step s
        f16x8 ah[2], al[2], wh[2][NB], wl[2][NB];
        auto frags = [&](int sidx, int set) {
            const int tc = sidx >> 1, ks = sidx & 1;
            const bool live = tc == 0 ? left_ok : tc == 2 ? right_ok : true;
            f16x8 h_ = *reinterpret_cast<const f16x8*>(As + x3_off(32 * wave + l31 + tc, 2 * ks + hi));
            f16x8 l_ = *reinterpret_cast<const f16x8*>(As + D2_A_PLANE + x3_off(32 * wave + l31 + tc, 2 * ks + hi));
            if (tc != 1) { h_ = live ? h_ : zero8; l_ = live ? l_ : zero8; }
            ah[set] = h_; al[set] = l_;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                wh[set][n] = *reinterpret_cast<const f16x8*>(Ws + (tc * 2) * X3_W_PLANE + x3_off(32 * n + l31, 2 * ks + hi));
                wl[set][n] = *reinterpret_cast<const f16x8*>(Ws + (tc * 2 + 1) * X3_W_PLANE + x3_off(32 * n + l31, 2 * ks + hi));
            }
        };
        frags(0, 0);
#pragma unroll
        for (int sidx = 0; sidx < 6; ++sidx) {
            const int set = sidx & 1;
            if (sidx + 1 < 6) frags(sidx + 1, set ^ 1);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[set][n], ah[set], acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[set][n], al[set], acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[set][n], ah[set], acc[n], 0, 0, 0);
        }
        __syncthreads();                 // step k + 1 landed (vmcnt(0)) and every wave is done reading stage cur
    }

    // ---- epilogue: as conv_igemm_x3_kernel<true>
    const long pp = p0 + 32 * wave + l31;
    if (pp >= P_total) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 bv[4 * NB], rv[4 * NB], sv[4 * NB];
    const bool with_res = p.out_mode == 0 && p.res;
    const float inv_a = 1.0f / p.in_scale;
    const long po = pp + (long)p.t_out_off * HoWo;
#pragma unroll
    for (int i = 0; i < 4 * NB; ++i) {
        const int co = co0 + 32 * (i >> 2) + 8 * (i & 3) + 4 * hi;
        const bool ok = co < p.Cout;
        if (NB == 1 && ok && co + 4 > p.Cout) {      // the partial last group of a narrow head: element-wise, pad channels read as zeros
            bv[i] = zero4; sv[i] = zero4; rv[i] = zero4;
            for (int e = 0; e < 4; ++e)
                if (co + e < p.Cout) {
                    if (p.bias) bv[i][e] = p.bias[co + e];
                    sv[i][e] = p.w2_inv[co + e] * inv_a;
                    if (with_res) rv[i][e] = p.res[po * p.ld_res + co + e];
                }
            continue;
        }
        bv[i] = (ok && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + co) : zero4;
        rv[i] = (ok && with_res) ? *reinterpret_cast<const f32x4*>(p.res + po * p.ld_res + co) : zero4;
        sv[i] = ok ? *reinterpret_cast<const f32x4*>(p.w2_inv + co) * inv_a : zero4;
    }
    long pq0 = 0;
    const int half = p.Cout >> 1;
    if (p.out_mode != 0) {
        const int t = (int)(pp / HoWo);
        const long sp = pp - (long)t * HoWo;
        pq0 = (long)(1 + 2 * (t - 1)) * HoWo + sp;
    }
#pragma unroll
    for (int i = 0; i < 4 * NB; ++i) {
        const int n = i >> 2, rg = i & 3;
        const int co = co0 + 32 * n + 8 * rg + 4 * hi;
        if (co >= p.Cout) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[n][4 * rg + e] * sv[i][e] + bv[i][e]) + rv[i][e];
        if (p.out_mode == 0) {
            *reinterpret_cast<f32x4*>(p.out + po * p.ld_out + co) = v;       // (narrow head: co + 4 <= ld_out, launch_conv_planes checks)
        } else {
            const int j = co >= half ? 1 : 0;
            *reinterpret_cast<f32x4*>(p.out + (pq0 + (long)j * HoWo) * p.ld_out + (co - j * half)) = v;
        }
    }
}

// =================================================================================================
// conv_dma2h_kernel<3> on TWO pixel tiles per workgroup that share every K step's weights (round 6).
// What the fill ablations say (tools/vae_ab.py on variant builds, profiles/r6i_vae_fill_ablation.txt, C2 decode 948 ms): without the weight pieces after
// the first step 674 ms, without the activation pieces 829 ms — the 36 KiB of weights every workgroup pulls per step (the same lines, for all 256 CUs at once)
// cost more than twice the 34 KiB of activations, and the matrix pipe is busy 47 % of the time because a step's 70 KiB take longer to arrive than its 54
// MFMAs per wave take to run.  Two tiles per workgroup halve the weight traffic per MFMA without more LDS: the weights are double-buffered as before
// (2 x 36 KiB), the two tiles' activation strips each have ONE buffer (2 x 34 KiB) and take turns — while tile 0's strip multiplies, tile 1's strip of
// the same step and half of the next step's weights stream in; while tile 1's multiplies, tile 0's strip of the NEXT step and the other half:
//   phase (k, 0)   MFMAs: tile 0 x W(k) from strip buffer 0      DMA: strip of tile 1, step k   -> buffer 1;  W(k+1) pieces 0..15  -> W[(k+1) & 1]
//   barrier        (strip buffer 1 and what landed are visible; buffer 0 is free)
//   phase (k, 1)   MFMAs: tile 1 x W(k) from strip buffer 1      DMA: strip of tile 0, step k+1 -> buffer 0;  W(k+1) pieces 16..35 -> W[(k+1) & 1]
//   barrier
// 52 KiB instead of 70 per 54 MFMAs and wave, the same 140 KiB of LDS, the same products in the same order per pixel: bit-identical to conv_dma2h_kernel<3>
// (SVI_VAE_PAIR=0 runs that one).  Measured (tools/vae_ab.py c2 ab-pair, same job): decode 957 -> 916 ms, encode 622 -> 609 ms — a quarter of what the byte
// count promised.  Tried on top, bit-identical, and dropped (profiles/r6i_vae_fill_ablation.txt): the SIMD's second wave placing its requests 27 MFMAs into
// the phase instead of at its start (-0.7 %); the pieces fetched into registers one phase ahead and written with ds_write_b128 (a whole phase more for them to
// arrive: +12 %, slower — two instructions per piece); every piece reading 1 KiB of contiguous memory, as a chunk-major plane layout would give (-8 % at most,
// wrong results: not built).  What the step waits for is therefore neither the bytes, nor the issue of the requests, nor their latency alone.
// =================================================================================================
#define D2P_W_STAGE (6 * X3_W_PLANE)             // 36 KiB: three x-taps x two planes x [96][64 B]
#define D2P_LDS (2 * D2P_W_STAGE + 2 * 2 * D2_A_PLANE)
__global__ __launch_bounds__(512, 2) void conv_dma2h_pair_kernel(ConvP p) {
    constexpr int NB = 3, WP = 2 * NB;
    constexpr int A_SLOTS = 5, W_SLOTS = 5;          // per wave: strip pieces q = wave + 8 i < 34, weight pieces w = wave + 8 i < 36
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Wbuf = smem;                         // W[0] | W[1]
    char* const Sbuf = smem + 2 * D2P_W_STAGE;       // strip buffer 0 | 1, each two planes of [272][64 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const long HoWo = (long)p.Ho * p.Wo;
    const long P_total = (long)p.To * HoWo;
    const int ncob = (p.Cout + NB * 32 - 1) / (NB * 32);
    const long pair = blockIdx.x / ncob;
    const int cob = (int)(blockIdx.x - pair * ncob);
    const int co0 = cob * NB * 32;
    const int nchunk = p.Cin >> 5;
    const int nrow = p.kt * p.kh;
    const int nk = nrow * nchunk;
    const unsigned OOB = 0xFFF00000u;

    long p0[2];
    int t_base = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        long tile = 2 * pair + s;                    // (a pair's second tile may lie past the last one: every pixel of it is masked out, nothing is stored)
        if (p.ord_T > 0) {
            const long per_group = (long)p.ord_T * p.ord_G;
            const long n_tiles = (long)p.ord_T * p.ord_Lf;
            if (tile < n_tiles) {
                const int full = p.ord_Lf / p.ord_G;
                const long g = tile / per_group;
                int gl = p.ord_G;
                long r = tile - g * per_group, gbase = g * p.ord_G;
                if (g >= full) { gl = p.ord_Lf - full * p.ord_G; r = tile - (long)full * per_group; gbase = (long)full * p.ord_G; }
                const int t = (int)(r / gl), j = (int)(r - (long)t * gl);
                tile = (long)t * p.ord_Lf + gbase + j;
            }
        }
        p0[s] = (long)p.t_begin * HoWo + tile * X3_PIX;
        const int t_first = (int)(max(min(p0[s] - 1, P_total - 1), 0L) / HoWo);
        t_base = min(t_base, max(t_first * p.st - p.pt, 0));
    }
    // ---- per-slot DMA bookkeeping
    unsigned a_off[2][A_SLOTS], a_mask[2][A_SLOTS], w_off[W_SLOTS];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int q = wave + 8 * i;
            a_off[s][i] = OOB; a_mask[s][i] = 0;
            if (q < 34) {
                const int r16 = q % 17, row = r16 * 16 + (lane >> 2);
                const long pp = p0[s] - 1 + row;
                const bool pix = pp >= 0 && pp < P_total;
                const long qq = pix ? pp : 0;
                const int at = (int)(qq / HoWo);
                const int rem = (int)(qq - (long)at * HoWo);
                const int ay = rem / p.Wo, ax = rem - ay * p.Wo;
                const int bt = at * p.st - p.pt, by = ay - p.ph;
                unsigned m = 0;
                for (int rt = 0; rt < nrow; ++rt) {
                    const int ta = rt / p.kh, tb = rt - ta * p.kh;
                    const int ti = bt + ta, yi = by + tb;
                    bool ok = pix && ti >= 0 && ti < p.Ti && yi >= 0 && yi < p.Hi;
                    if (p.zero_frame0 && ti == 0) ok = false;
                    if (ok) m |= 1u << rt;
                }
                a_mask[s][i] = m;
                const int chunk = (lane & 3) ^ ((row >> 2) & 3);
                a_off[s][i] = (unsigned)((((bt - t_base) * p.Hi + by) * p.Wi + ax) * p.ld_in) * 2u + (unsigned)chunk * 16u;
            }
        }
#pragma unroll
    for (int i = 0; i < W_SLOTS; ++i) {
        const int w = wave + 8 * i;
        w_off[i] = OOB;
        if (w < 6 * WP) {
            const int tc = w / (2 * WP), pl = (w % (2 * WP)) / WP, row = (w % WP) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            if (co0 + row < p.Cout) w_off[i] = (unsigned)((pl * p.plane_w3 + ((long)tc * p.Cout + co0 + row) * p.ld_w3 + chunk * 8) * 2);
        }
    }
    const long base_el = (long)t_base * p.Hi * p.Wi * p.ld_in;
    const long rem_bytes = ((long)p.Ti * p.Hi * p.Wi * p.ld_in - base_el) * 2;
    const int win = (int)(unsigned)min(rem_bytes, 0xFFE00000L);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.in_h + base_el), 0, win, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.in_l + base_el), 0, win, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2h), 0, (int)(unsigned)min((long)2 * p.plane_w3 * 2, 0xFFE00000L), 0x00020000);
    bool left_ok[2], right_ok[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const long pp = min(p0[s] + 32 * wave + l31, P_total - 1);
        const int ax = (int)(pp % p.Wo);
        left_ok[s] = ax > 0; right_ok[s] = ax < p.Wo - 1;
    }

    // K step iterators: `c*` = the step being multiplied, `n*` = the step after it
    int c_rt = 0, c_cc = 0, c_ta = 0, c_tb = 0, n_rt = 0, n_cc = 0, n_ta = 0, n_tb = 0;
    auto step_next = [&](int& rt, int& cc, int& ta, int& tb) {
        if (++cc == nchunk) {
            cc = 0; ++rt;
            if (++tb == p.kh) { tb = 0; ++ta; }
        }
    };
    auto request_strip = [&](auto sc, int rt, int cc, int ta, int tb) {          // tile S's strip of step (rt, cc) -> strip buffer S
        constexpr int S = decltype(sc)::value;
        char* As = Sbuf + S * 2 * D2_A_PLANE;
        const unsigned d_act = (unsigned)(((ta * p.Hi + tb) * p.Wi) * p.ld_in) * 2u + (unsigned)cc * 64u;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int q = wave + 8 * i;
            if (q < 34) {
                const int pl = q / 17, r16 = q % 17;
                const unsigned off = ((a_mask[S][i] >> rt) & 1u) ? a_off[S][i] + d_act : OOB;
                if (pl == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (lptr_t)(As + r16 * 1024), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, (lptr_t)(As + D2_A_PLANE + r16 * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    auto request_w = [&](int half, int rt, int cc, int wb) {                    // weight pieces [0, 16) (half 0) or [16, 36) (half 1) of step (rt, cc) -> W[wb]
        char* Ws = Wbuf + wb * D2P_W_STAGE;
        const unsigned d_w = (unsigned)((long)rt * p.kw * p.Cout * p.ld_w3 * 2) + (unsigned)cc * 64u;
#pragma unroll
        for (int i = 0; i < W_SLOTS; ++i) {
            const int w = wave + 8 * i;
            if ((i < 2) != (half == 0) || w >= 6 * WP) continue;
            const int tc = w / (2 * WP), pl = (w % (2 * WP)) / WP, r16 = w % WP;
            const unsigned off = w_off[i] == OOB ? OOB : w_off[i] + d_w;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(Ws + (tc * 2 + pl) * X3_W_PLANE + r16 * 1024), 16, off, 0, 0, 0);
        }
    };

    f32x16 acc[2][NB];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][n][r] = 0.f;
    const f16x8 zero8 = {(f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0};
    // the 54 MFMAs of one (tile, step): exactly conv_dma2h_kernel's six sub-steps
    auto multiply = [&](auto sc, int wb) {
        constexpr int S = decltype(sc)::value;
        const char* As = Sbuf + S * 2 * D2_A_PLANE;
        const char* Ws = Wbuf + wb * D2P_W_STAGE;
        f16x8 ah[2], al[2], wh[2][NB], wl[2][NB];
        auto frags = [&](int sidx, int set) {
            const int tc = sidx >> 1, ks = sidx & 1;
            const bool live = tc == 0 ? left_ok[S] : tc == 2 ? right_ok[S] : true;
            f16x8 h_ = *reinterpret_cast<const f16x8*>(As + x3_off(32 * wave + l31 + tc, 2 * ks + hi));
            f16x8 l_ = *reinterpret_cast<const f16x8*>(As + D2_A_PLANE + x3_off(32 * wave + l31 + tc, 2 * ks + hi));
            if (tc != 1) { h_ = live ? h_ : zero8; l_ = live ? l_ : zero8; }
            ah[set] = h_; al[set] = l_;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                wh[set][n] = *reinterpret_cast<const f16x8*>(Ws + (tc * 2) * X3_W_PLANE + x3_off(32 * n + l31, 2 * ks + hi));
                wl[set][n] = *reinterpret_cast<const f16x8*>(Ws + (tc * 2 + 1) * X3_W_PLANE + x3_off(32 * n + l31, 2 * ks + hi));
            }
        };
        frags(0, 0);
#pragma unroll
        for (int sidx = 0; sidx < 6; ++sidx) {
            const int set = sidx & 1;
            if (sidx + 1 < 6) frags(sidx + 1, set ^ 1);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[S][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[set][n], ah[set], acc[S][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[S][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[set][n], al[set], acc[S][n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[S][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[set][n], ah[set], acc[S][n], 0, 0, 0);
        }
    };
    std::integral_constant<int, 0> T0;
    std::integral_constant<int, 1> T1;

    request_w(0, 0, 0, 0);
    request_w(1, 0, 0, 0);
    request_strip(T0, 0, 0, 0, 0);
    step_next(n_rt, n_cc, n_ta, n_tb);               // n* = step 1
    __syncthreads();                                 // (hipcc drains the LDS-DMA with vmcnt(0) in front of the barrier)
    for (int k = 0; k < nk; ++k) {
        const int wb = k & 1;
        const bool more = k + 1 < nk;
        request_strip(T1, c_rt, c_cc, c_ta, c_tb);
        if (more) request_w(0, n_rt, n_cc, wb ^ 1);
        multiply(T0, wb);
        __syncthreads();                             // tile 1's strip and the first weight pieces have landed; every wave is done with strip buffer 0
        if (more) {
            request_strip(T0, n_rt, n_cc, n_ta, n_tb);
            request_w(1, n_rt, n_cc, wb ^ 1);
        }
        multiply(T1, wb);
        __syncthreads();                             // tile 0's next strip and W(k+1) have landed; every wave is done with strip buffer 1 and W[wb]
        step_next(c_rt, c_cc, c_ta, c_tb);
        step_next(n_rt, n_cc, n_ta, n_tb);
    }

    // ---- epilogue: conv_dma2h_kernel's, once per tile
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool with_res = p.out_mode == 0 && p.res;
    const float inv_a = 1.0f / p.in_scale;
    const int half = p.Cout >> 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const long pp = p0[s] + 32 * wave + l31;
        if (pp >= P_total) continue;
        const long po = pp + (long)p.t_out_off * HoWo;
        long pq0 = 0;
        if (p.out_mode != 0) {
            const int t = (int)(pp / HoWo);
            const long sp = pp - (long)t * HoWo;
            pq0 = (long)(1 + 2 * (t - 1)) * HoWo + sp;
        }
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) {
            const int n = i >> 2, rg = i & 3;
            const int co = co0 + 32 * n + 8 * rg + 4 * hi;
            if (co >= p.Cout) continue;
            const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co) : zero4;
            const f32x4 rv = with_res ? *reinterpret_cast<const f32x4*>(p.res + po * p.ld_res + co) : zero4;
            const f32x4 sv = *reinterpret_cast<const f32x4*>(p.w2_inv + co) * inv_a;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc[s][n][4 * rg + e] * sv[e] + bv[e]) + rv[e];
            if (p.out_mode == 0) {
                *reinterpret_cast<f32x4*>(p.out + po * p.ld_out + co) = v;
            } else {
                const int j = co >= half ? 1 : 0;
                *reinterpret_cast<f32x4*>(p.out + (pq0 + (long)j * HoWo) * p.ld_out + (co - j * half)) = v;
            }
        }
    }
}

// Can a convolution take its input as the producer's two fp16 planes (conv_dma2h_kernel)?  Decided BEFORE the producer runs.
bool conv_planes_ok(const ConvP& p) {
    const bool vec_ok = (((uintptr_t)p.w2_inv | (uintptr_t)p.bias) & 15) == 0;
    const bool wide = p.Cout >= 64 && p.Cout % 4 == 0 && vec_ok && (p.out_mode == 0 || (p.Cout / 2) % 4 == 0);
    const bool narrow = p.Cout <= 32 && p.out_mode == 0 && (p.Cout % 4 != 0 || vec_ok);          // one 32-channel block; any channel count (conv_dma2h_kernel<1>)
    return p.w2h && p.w2_inv && p.in_scale > 0.f && !svi_switches().vae_no_x2h && !svi_switches().vae_exact_fp32 && svi_switches().vae_dma &&
           p.Cin % 32 == 0 && p.ld_in == p.Cin && !p.ups && (wide || narrow) &&
           p.kw == 3 && p.kh == 3 && p.kt * p.kh <= 32 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.ph == 1 && p.pw == 1 && p.Ho == p.Hi && p.Wo == p.Wi &&
           !p.act_silu && !p.out_bf16 &&
           (long)(p.kt + 3) * p.Hi * p.Wi * p.ld_in * 2 < 0xFFE00000L && (long)2 * p.plane_w3 * 2 < 0xFFE00000L;
}

svi_status launch_conv_planes(const ConvP& p, hipStream_t st) {
    SVI_REQUIRE(conv_planes_ok(p) && p.in_h && p.in_l, "conv: this layer cannot take fp16 input planes");
    SVI_REQUIRE(p.ld_out % 4 == 0 && (!p.res || p.ld_res % 4 == 0) && (((uintptr_t)p.res | (uintptr_t)p.out) & 15) == 0, "conv: output / residual alignment");
    const long pixels = (long)(p.To - p.t_begin) * p.Ho * p.Wo;
    if (pixels <= 0) return SVI_OK;
    ConvP q = p;
    {   // frame-interleaved tile order (see the kernel) when frames are whole numbers of tiles and there is more than one
        const long hw = (long)p.Ho * p.Wo;
        const int frames = p.To - p.t_begin;
        q.ord_T = 0;
        if (svi_switches().vae_tile_order && p.kt > 1 && frames > 1 && hw % X3_PIX == 0) {
            q.ord_T = frames; q.ord_Lf = (int)(hw / X3_PIX);
            q.ord_G = (int)std::max<long>(1, std::min<long>(q.ord_Lf, (16L * p.Wo + X3_PIX - 1) / X3_PIX));
        }
    }
    if (p.Cout <= 32) {                  // narrow head
        SVI_REQUIRE((p.Cout + 3) / 4 * 4 <= p.ld_out && (!p.res || (p.Cout + 3) / 4 * 4 <= p.ld_res), "conv: a narrow head writes whole groups of four channels (Cout=%d, ld_out=%d)", p.Cout, p.ld_out);
        SVI_REQUIRE(p.Cout % 4 != 0 || (((uintptr_t)p.bias | (uintptr_t)p.w2_inv) & 15) == 0, "conv: bias / scale alignment");
        dim3 grid((unsigned)((pixels + X3_PIX - 1) / X3_PIX), 1), block(512);
        SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_dma2h_kernel<1>), 2 * D2_STAGE));
        hipLaunchKernelGGL(conv_dma2h_kernel<1>, grid, block, 2 * D2_STAGE, st, q);
    } else {
        SVI_REQUIRE((((uintptr_t)p.bias) & 15) == 0, "conv: bias alignment");
        const long nwg = ((pixels + X3_PIX - 1) / X3_PIX) * ((p.Cout + X3_CO - 1) / X3_CO);
        SVI_REQUIRE(nwg < (1L << 31), "conv: too many workgroups");
        dim3 grid((unsigned)nwg, 1), block(512);
        if (svi_switches().vae_pair) {       // two tiles per workgroup sharing every step's weights (conv_dma2h_pair_kernel)
            const long ntile = (pixels + X3_PIX - 1) / X3_PIX;
            dim3 gridp((unsigned)(((ntile + 1) / 2) * ((p.Cout + X3_CO - 1) / X3_CO)), 1);
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_dma2h_pair_kernel), D2P_LDS));
            hipLaunchKernelGGL(conv_dma2h_pair_kernel, gridp, block, D2P_LDS, st, q);
            SVI_LAUNCH_CHECK();
            return SVI_OK;
        }
        SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_dma2h_kernel<3>), 2 * D2_STAGE));
        hipLaunchKernelGGL(conv_dma2h_kernel<3>, grid, block, 2 * D2_STAGE, st, q);
    }
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// Does conv_igemm_x3_kernel take this convolution?
bool conv_x3_ok(const ConvP& p) {
    const bool no_x3 = svi_switches().vae_exact_fp32 != 0;                 // A/B aid: force the exact-fp32 MFMA kernel
    return p.w3 && !no_x3 && !p.act_silu && !p.out_bf16 && p.Cout >= 64 && p.Cout % 4 == 0 && p.ld_out % 4 == 0 && (!p.res || p.ld_res % 4 == 0) &&
           (p.out_mode == 0 || (p.Cout / 2) % 4 == 0) && (((uintptr_t)p.bias | (uintptr_t)p.res | (uintptr_t)p.out) & 15) == 0 &&
           p.kt * p.kh * p.kw <= 32 &&                                                   // tap bit mask
           (long)(p.kt + 3) * p.Hi * p.Wi * p.ld_in * 4 < 0xFFFFF000L &&               // 32-bit offsets inside the buffer window
           (long)3 * p.plane_w3 * 2 < 0xFFFFF000L;
}

svi_status launch_conv(const ConvP& p, hipStream_t st) {
    SVI_REQUIRE(p.Cin % 4 == 0 && p.ld_in % 4 == 0 && p.ld_w % 4 == 0, "conv: Cin/ld must be multiples of 4 (Cin=%d)", p.Cin);
    SVI_REQUIRE(!p.up_phase || (conv_x3_ok(p) && !p.ups && p.out_mode == 0), "conv: an upsample phase runs on the three-term kernel only");
    const long pixels = (long)(p.To - p.t_begin) * p.Ho * p.Wo;
    if (pixels <= 0) return SVI_OK;
    if (conv_x3_ok(p)) {
        dim3 grid3((unsigned)((pixels + X3_PIX - 1) / X3_PIX), (unsigned)((p.Cout + X3_CO - 1) / X3_CO)), block3(512);
        ConvP pa = p;
#ifdef SVI_ABLATIONS
        pa.abl = svi_switches().vae_abl;
#endif
        if (p.w2h && p.w2_inv && p.in_scale > 0.f && !svi_switches().vae_no_x2h && (((uintptr_t)p.w2_inv) & 15) == 0 && p.Cin % 32 == 0 && !p.ups && !p.up_phase &&
            (long)(p.kt + 3) * p.Hi * p.Wi * p.ld_in * 4 < 0xFFE00000L && (long)2 * p.plane_w3 * 2 < 0xFFE00000L) {
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_igemm_x3_kernel<true>), 2 * X2H_STAGE));
            hipLaunchKernelGGL(conv_igemm_x3_kernel<true>, grid3, block3, 2 * X2H_STAGE, st, pa);
        } else {
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_igemm_x3_kernel<false>), 2 * X3_STAGE));
            hipLaunchKernelGGL(conv_igemm_x3_kernel<false>, grid3, block3, 2 * X3_STAGE, st, pa);
        }
        SVI_LAUNCH_CHECK();
        return SVI_OK;
    }
    const int NT = p.Cout > 32 ? 3 : 1;
    dim3 grid((unsigned)((pixels + 127) / 128), (unsigned)((p.Cout + 32 * NT - 1) / (32 * NT))), block(256);
    SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_igemm_kernel<3>), 2 * (128 * 128 + 3 * 32 * 128)));
    SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(conv_igemm_kernel<1>), 2 * (128 * 128 + 1 * 32 * 128)));
    if (NT == 3)
        hipLaunchKernelGGL(conv_igemm_kernel<3>, grid, block, 2 * (128 * 128 + 3 * 32 * 128), st, p);
    else
        hipLaunchKernelGGL(conv_igemm_kernel<1>, grid, block, 2 * (128 * 128 + 1 * 32 * 128), st, p);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// A 3x3 (spatial) convolution behind a nearest x2 upsample (Resample 'upsample2d/3d', vae:120-131), as FOUR 2x2 convolutions of the
// small image: output pixel (2 y + py, 2 x + px) sees only two distinct input rows and two distinct input columns —
//     py = 0: rows y-1 (kernel row 0) and y (kernel rows 1 + 2);     py = 1: rows y (kernel rows 0 + 1) and y+1 (kernel row 2)
// (columns alike), and positions outside the small image are exactly the positions whose upsampled pixels fall into the convolution's
// zero padding.  Summing the kernel entries that meet the same input pixel beforehand (fp32, svi_vae_set_weight) leaves 4 taps of
// the 9: 2.25 x fewer products for the same sum, up to the rounding of the weight sums (<= 2 ulp of fp32 per entry; the three-term
// operand split the kernel runs on is itself good to 2^-24).  p describes the SMALL grid (ups = 0, kh = kw = 2); w3_up holds the
// four phases' weights, each in the three-plane layout with 4 taps.
svi_status launch_conv_up_phases(ConvP p, const bf16* w3_up, hipStream_t st) {
    p.ups = 0; p.kh = p.kw = 2;
    p.Ho = p.Hi; p.Wo = p.Wi;
    p.plane_w3 = (long)p.kt * 4 * p.Cout * p.ld_w3;
    p.w2h = nullptr; p.w = nullptr;
    for (int ph = 0; ph < 4; ++ph) {
        p.up_phase = 1 + ph;
        p.ph = 1 - (ph >> 1); p.pw = 1 - (ph & 1);
        p.w3 = w3_up + (long)ph * 3 * p.plane_w3;
        SVI_TRY(launch_conv(p, st));
    }
    return SVI_OK;
}

}  // namespace

// fp32 linear layer C = A W^T + bias (+ res) on the exact-fp32 MFMA kernel (a 1x1x1 "convolution": pixels = rows).  Used by the CLIP
// image encoder, which the reference runs in fp32 (pipelines/svi_video.py:307-309).
svi_status svi_launch_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* Cm, int ldc, int M, int N, int K,
                               const float* res, int ldres, hipStream_t st) {
    SVI_REQUIRE(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "gemm_f32: K and the leading dimensions must be multiples of 4");
    SVI_REQUIRE(lda >= K && ldw >= K && ldc >= N && (!res || ldres >= N), "gemm_f32: leading dimensions too small");
    SVI_REQUIRE((((uintptr_t)A | (uintptr_t)W) & 15) == 0, "gemm_f32: operands must be 16-byte aligned");
    if (M == 0) return SVI_OK;
    ConvP g{};
    g.in = A; g.Ti = 1; g.Hi = 1; g.Wi = M; g.Cin = K; g.ld_in = lda;
    g.w = W; g.ld_w = ldw; g.bias = bias;
    g.out = Cm; g.To = 1; g.Ho = 1; g.Wo = M; g.Cout = N; g.ld_out = ldc;
    g.kt = g.kh = g.kw = 1; g.st = g.sh = g.sw = 1;
    g.res = res; g.ld_res = ldres;
    return launch_conv(g, st);
}

namespace {
// ---- RMS_norm over channels (+SiLU):  x / max(||x||_2, 1e-12) * sqrt(C) * gamma   (vae:55-70, 207-209) ----
// Eight lanes per pixel, 16 bytes per lane and access: lane q of a pixel owns channels 4 q + 32 i .. + 3, so every wave instruction moves
// eight whole 128-byte segments (was: 32 lanes per pixel with 4-byte accesses, 3.4 TB/s on the 12.4 GB layers).  C % 4 == 0.
// Arithmetic per element (both kernels, one function): the pixel's factor sqrt(C) / max(||x||, eps) is formed once (IEEE division), then
// y = x * factor * gamma and SiLU as y * rcp(1 + exp2(-y log2 e)) on the hardware's v_exp / v_rcp (1 ulp each).  (Was: two IEEE divisions
// and an expf per element, ~40 VALU instructions against 8 bytes of traffic — the 12.4 GB layers ran at 4.4 TB/s, arithmetic-bound.)
__device__ __forceinline__ float norm_silu_el(float x, float factor, float g, int do_silu) {
    const float y = x * factor * g;
    if (!do_silu) return y;
    const float e = __builtin_amdgcn_exp2f(y * -1.4426950408889634f);
    return y * __builtin_amdgcn_rcpf(1.0f + e);
}
template <int MAXI>
__global__ __launch_bounds__(256) void rms_silu_kernel(const float* __restrict__ in, float* __restrict__ out, long pixels,
                                                       int C, const float* __restrict__ gamma, int do_silu) {
    const long px = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int q = threadIdx.x & 7;
    if (px >= pixels) return;                   // whole 8-lane groups leave together: the shuffles below stay inside a group
    const float* ip = in + px * C;
    f32x4 v[MAXI];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = 4 * q + 32 * i;
        if (c < C) v[i] = *reinterpret_cast<const f32x4*>(ip + c);
        else v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ss += v[i][0] * v[i][0];
        ss += v[i][1] * v[i][1];
        ss += v[i][2] * v[i][2];
        ss += v[i][3] * v[i][3];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float factor = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
    float* op = out + px * C;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = 4 * q + 32 * i;
        if (c < C) {
            const float g[4] = {gamma[c], gamma[c + 1], gamma[c + 2], gamma[c + 3]};      // borrowed parameter: no alignment promise beyond 4 bytes
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = norm_silu_el(v[i][e], factor, g[e], do_silu);
            *reinterpret_cast<f32x4*>(op + c) = y;
        }
    }
}

svi_status launch_rms_silu(const float* in, float* out, long pixels, int C, const float* gamma, int do_silu, hipStream_t st) {
    SVI_REQUIRE(C <= 384 && C % 4 == 0, "vae rms norm: C=%d (need a multiple of 4, at most 384)", C);
    SVI_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "vae rms norm: activations must be 16-byte aligned");
    if (pixels <= 0) return SVI_OK;
    dim3 grid((unsigned)((pixels + 31) / 32)), block(256);
    if (C <= 96) hipLaunchKernelGGL(rms_silu_kernel<3>, grid, block, 0, st, in, out, pixels, C, gamma, do_silu);
    else if (C <= 192) hipLaunchKernelGGL(rms_silu_kernel<6>, grid, block, 0, st, in, out, pixels, C, gamma, do_silu);
    else hipLaunchKernelGGL(rms_silu_kernel<12>, grid, block, 0, st, in, out, pixels, C, gamma, do_silu);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// The same RMS_norm (+ SiLU), written as the two fp16 words the two-term convolution multiplies: hi = f16(y s), lo = f16(y s - hi) with the
// power-of-two scale s the bound of y allows (|y| s <= 2^15) — exactly the split conv_igemm_x3_kernel<true> performs on the fly, done ONCE by
// the producer.  Planes are channels-last [pixel][C] fp16; together they are as large as the fp32 tensor they replace.
// Eight lanes per pixel; lane q owns channels 8 q + 64 i .. + 7 (two 16-byte loads, ONE 16-byte store per plane).  C % 8 == 0.
template <int MAXI>
__global__ __launch_bounds__(256) void rms_silu_planes_kernel(const float* __restrict__ in, unsigned short* __restrict__ out_h, unsigned short* __restrict__ out_l,
                                                              long pixels, int C, const float* __restrict__ gamma, int do_silu, float s) {
    const long px = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int q = threadIdx.x & 7;
    if (px >= pixels) return;
    const float* ip = in + px * C;
    f32x4 v[MAXI][2];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = 8 * q + 64 * i;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (c < C) v[i][hf] = *reinterpret_cast<const f32x4*>(ip + c + 4 * hf);
            else v[i][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
            ss += v[i][hf][0] * v[i][hf][0];
            ss += v[i][hf][1] * v[i][hf][1];
            ss += v[i][hf][2] * v[i][hf][2];
            ss += v[i][hf][3] * v[i][hf][3];
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float factor = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = 8 * q + 64 * i;
        if (c < C) {
            u16x8 hh, ll;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = norm_silu_el(v[i][e >> 2][e & 3], factor, gamma[c + e], do_silu);
                const f16 hw = (f16)(t * s);
                hh[e] = bits16(hw);
                ll[e] = bits16((f16)__builtin_fmaf(t, s, -(float)hw));
            }
            *reinterpret_cast<u16x8*>(out_h + px * C + c) = hh;
            *reinterpret_cast<u16x8*>(out_l + px * C + c) = ll;
        }
    }
}

svi_status launch_rms_silu_planes(const float* in, unsigned short* out_h, unsigned short* out_l, long pixels, int C, const float* gamma, int do_silu, float s,
                                  hipStream_t st) {
    SVI_REQUIRE(C <= 384 && C % 8 == 0 && s > 0.f, "vae rms norm (planes): C=%d", C);
    SVI_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out_h % 16) == 0 && ((uintptr_t)out_l % 16) == 0, "vae rms norm (planes): alignment");
    if (pixels <= 0) return SVI_OK;
    dim3 grid((unsigned)((pixels + 31) / 32)), block(256);
    if (C <= 128) hipLaunchKernelGGL(rms_silu_planes_kernel<2>, grid, block, 0, st, in, out_h, out_l, pixels, C, gamma, do_silu, s);
    else if (C <= 192) hipLaunchKernelGGL(rms_silu_planes_kernel<3>, grid, block, 0, st, in, out_h, out_l, pixels, C, gamma, do_silu, s);
    else hipLaunchKernelGGL(rms_silu_planes_kernel<6>, grid, block, 0, st, in, out_h, out_l, pixels, C, gamma, do_silu, s);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ---- row softmax with scale (AttentionBlock, vae:262) ---------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int rows, int cols, int ld, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = s + (long)row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, r[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf((r[c] - mx) * scale);
        r[c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < cols; c += 64) r[c] *= inv;
}

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out,
                                                            int ldo, int rows, int cols) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ldi + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(long)c * ldo + r] = tile[tx][i];
    }
}

// ---- layout / (de)normalisation kernels ---------------------------------------------------------------------------
// latents [16,T,h,w] -> channels-last [T,h,w,16] with z / (1/std) + mean (vae:555-561)
__global__ void latent_in_kernel(const float* __restrict__ z, float* __restrict__ out, long thw, const float* __restrict__ mean,
                                 const float* __restrict__ inv_std) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= thw * 16) return;
    const int c = (int)(i & 15);
    const long sp = i >> 4;
    out[i] = z[(long)c * thw + sp] / inv_std[c] + mean[c];
}
// channels-last mu [T,h,w,ld] -> latents [16,T,h,w] with (mu - mean) * (1/std) (vae:542-549)
__global__ void latent_out_kernel(const float* __restrict__ mu, int ld, float* __restrict__ z, long thw, const float* __restrict__ mean,
                                  const float* __restrict__ inv_std) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= thw * 16) return;
    const int c = (int)(i / thw);
    const long sp = i - (long)c * thw;
    z[i] = (mu[sp * ld + c] - mean[c]) * inv_std[c];
}
// video [3,T,H,W] -> channels-last [T,H,W,4] (4th channel zero)
__global__ void video_in_kernel(const float* __restrict__ v, float* __restrict__ out, long thw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= thw * 4) return;
    const int c = (int)(i & 3);
    out[i] = c < 3 ? v[(long)c * thw + (i >> 2)] : 0.f;
}
// channels-last [T,H,W,ld] -> video [3,T,H,W], clamped to [-1,1] (vae:753-756)
__global__ void video_out_kernel(const float* __restrict__ in, int ld, float* __restrict__ v, long thw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= thw * 3) return;
    const int c = (int)(i / thw);
    const long sp = i - (long)c * thw;
    v[i] = fminf(fmaxf(in[sp * ld + c], -1.f), 1.f);
}
// ---- spatial tiling (WanVideoVAE.tiled_decode / tiled_encode, vae:643-744) -------------------------------------------
// A tile is a window [h0, h0+th) x [w0, w0+tw) of the caller's [C, T, Hf, Wf] tensor: read in place (no gathered copy), its
// result multiplied by the reference's linear-ramp mask and accumulated into the full-size output; `weight` accumulates the mask
// ([Hf, Wf] plane: the reference's [1,1,T,H,W] weight does not depend on t), the finalisation divides (and, for decode, clamps).
// Every product / sum / quotient is a separately rounded fp32 operation in the reference's order (tiles in task order), so the
// blend itself is bit-identical to the reference's CPU arithmetic given the same tile values.
struct TileWin {
    int Hf, Wf, h0, w0, th, tw;          // full plane, window origin and size (in elements of the tensor the window is cut from)
    int oHf, oWf, oh0, ow0;              // the same window in the output tensor's resolution
    int lb, rb, tb, bb;                  // is_bound: (top, bottom, left, right) = (h==0, h_>=H, w==0, w_>=W)   vae:663
    int border_h, border_w;              // ramp widths in output elements
};
// build_1d_mask (vae:621-627): ones, ramp (i+1)/border on a non-bound left edge, mirrored on a non-bound right edge (the right
// edge is written last, so it wins where the two ramps overlap)
__device__ __forceinline__ float mask_1d(int i, int length, int left_bound, int right_bound, int border) {
    float m = 1.f;
    if (!left_bound && i < border) m = (float)(i + 1) / (float)border;
    if (!right_bound && i >= length - border) m = (float)(length - i) / (float)border;
    return m;
}
__global__ void latent_in_win_kernel(const float* __restrict__ z, float* __restrict__ out, int T, TileWin tw, const float* __restrict__ mean,
                                     const float* __restrict__ inv_std) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)T * tw.th * tw.tw * 16;
    if (i >= n) return;
    const int c = (int)(i & 15);
    long sp = i >> 4;
    const int x = (int)(sp % tw.tw); sp /= tw.tw;
    const int y = (int)(sp % tw.th);
    const int t = (int)(sp / tw.th);
    out[i] = z[(((long)c * T + t) * tw.Hf + tw.h0 + y) * tw.Wf + tw.w0 + x] / inv_std[c] + mean[c];
}
__global__ void video_in_win_kernel(const float* __restrict__ v, float* __restrict__ out, int T, TileWin tw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)T * tw.th * tw.tw * 4;
    if (i >= n) return;
    const int c = (int)(i & 3);
    long sp = i >> 2;
    const int x = (int)(sp % tw.tw); sp /= tw.tw;
    const int y = (int)(sp % tw.th);
    const int t = (int)(sp / tw.th);
    out[i] = c < 3 ? v[(((long)c * T + t) * tw.Hf + tw.h0 + y) * tw.Wf + tw.w0 + x] : 0.f;
}
// tile result, channels-last [To, oth, otw, ld] -> values[C][To][oHf][oWf] += v * mask ; weight[oHf][oWf] += mask   (vae:668-685)
// mode 0: decode (v = raw decoder output, NOT clamped: the reference clamps after the blend, vae:687);
// mode 1: encode (v = (mu - mean) * inv_std, vae:542-549)
__global__ void tile_blend_kernel(const float* __restrict__ in, int ld, float* __restrict__ values, float* __restrict__ weight, int C, int To,
                                  int oth, int otw, TileWin tw, int mode, const float* __restrict__ mean, const float* __restrict__ inv_std) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)To * oth * otw;
    if (i >= n) return;
    const int x = (int)(i % otw);
    const int y = (int)((i / otw) % oth);
    const int t = (int)(i / ((long)otw * oth));
    const float m = fminf(mask_1d(y, oth, tw.lb, tw.rb, tw.border_h), mask_1d(x, otw, tw.tb, tw.bb, tw.border_w));
    const long pix = (long)(tw.oh0 + y) * tw.oWf + tw.ow0 + x;
    const long plane = (long)tw.oHf * tw.oWf;
    for (int c = 0; c < C; ++c) {
        float v = in[i * ld + c];
        if (mode == 1) v = (v - mean[c]) * inv_std[c];
        const long o = ((long)c * To + t) * plane + pix;
        values[o] = values[o] + v * m;
    }
    if (t == 0) weight[pix] = weight[pix] + m;
}
__global__ void tile_finalize_kernel(float* __restrict__ values, const float* __restrict__ weight, long planes, long plane, int clamp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * plane) return;
    float v = values[i] / weight[i % plane];
    if (clamp) v = fminf(fmaxf(v, -1.f), 1.f);
    values[i] = v;
}
// weights [Cout, Cin, kt, kh, kw] -> [tap][Cout][ldw] (ldw = Cin rounded up to 4, zero padded)
// weights [Cout, Cin, taps] -> three bf16 planes [plane][tap][Cout][ldw3] with w = hi + mid + lo (ldw3 = Cin rounded up to 32)
__global__ void pack_weight_x3_kernel(const float* __restrict__ w, bf16* __restrict__ out, int Cout, int Cin, int taps, int ldw3) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)taps * Cout * ldw3;
    if (i >= n) return;
    const int ci = (int)(i % ldw3);
    const int co = (int)((i / ldw3) % Cout);
    const int tap = (int)(i / ((long)ldw3 * Cout));
    const float x = ci < Cin ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
    const bf16 h = (bf16)x;
    const float r1 = x - (float)h;
    const bf16 m = (bf16)r1;
    out[i] = h;
    out[n + i] = m;
    out[2 * n + i] = (bf16)(r1 - (float)m);
}
// weights [Cout, Cin, 3, 3] of a convolution behind a nearest x2 upsample -> four phases x three bf16 planes [phase][plane][2x2 tap][Cout][ldw3]
// of the summed kernels (launch_conv_up_phases).  Sums in fp32, kernel rows outer, columns inner.
__global__ void pack_weight_up_x3_kernel(const float* __restrict__ w, bf16* __restrict__ out, int Cout, int Cin, int ldw3) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)4 * Cout * ldw3;            // one plane of one phase
    if (i >= 4 * n) return;
    const int ph = (int)(i / n);
    const long r = i - (long)ph * n;
    const int ci = (int)(r % ldw3);
    const int co = (int)((r / ldw3) % Cout);
    const int tap = (int)(r / ((long)ldw3 * Cout));
    const int py = ph >> 1, px = ph & 1, a = tap >> 1, b = tap & 1;
    // kernel rows met by tap row a of phase py: py = 0: {0} | {1, 2};  py = 1: {0, 1} | {2}
    const int r0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), r1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int c0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), c1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float x = 0.f;
    if (ci < Cin)
        for (int rr = r0; rr <= r1; ++rr)
            for (int cc = c0; cc <= c1; ++cc) x += w[((long)co * Cin + ci) * 9 + rr * 3 + cc];
    const bf16 h = (bf16)x;
    const float r1f = x - (float)h;
    const bf16 m = (bf16)r1f;
    bf16* o = out + (long)ph * 3 * n + r;
    o[0] = h;
    o[n] = m;
    o[2 * n] = (bf16)(r1f - (float)m);
}
// fp16 two-term form.  Row (output channel) maximum -> power-of-two scale 2^k with max * 2^k in [2^13, 2^14); inv[co] = 2^-k.
__global__ __launch_bounds__(256) void weight_row_scale_kernel(const float* __restrict__ w, float* __restrict__ scale, float* __restrict__ inv, long per_row) {
    const int co = blockIdx.x;
    float m = 0.f;
    for (long i = threadIdx.x; i < per_row; i += 256) m = fmaxf(m, fabsf(w[(long)co * per_row + i]));
    m = wave_max(m);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        int e = 0;
        if (m > 0.f && m < INFINITY) (void)frexpf(m, &e);          // m = f * 2^e, f in [0.5, 1)
        else e = 14;
        e = min(max(e, -100), 100);
        scale[co] = ldexpf(1.0f, 14 - e);
        inv[co] = ldexpf(1.0f, e - 14);
    }
}
__global__ void pack_weight_x2h_kernel(const float* __restrict__ w, const float* __restrict__ scale, unsigned short* __restrict__ out, int Cout,
                                       int Cin, int taps, int ldw3) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)taps * Cout * ldw3;
    if (i >= n) return;
    const int ci = (int)(i % ldw3);
    const int co = (int)((i / ldw3) % Cout);
    const int tap = (int)(i / ((long)ldw3 * Cout));
    const float x = ci < Cin ? w[((long)co * Cin + ci) * taps + tap] * scale[co] : 0.f;
    const f16 h = (f16)x;
    out[i] = bits16(h);
    out[n + i] = bits16((f16)(x - (float)h));
}
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int taps, int ldw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)taps * Cout * ldw;
    if (i >= n) return;
    const int ci = (int)(i % ldw);
    const int co = (int)((i / ldw) % Cout);
    const int tap = (int)(i / ((long)ldw * Cout));
    out[i] = ci < Cin ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
}

const float kMean[16] = {-0.7571f, -0.7089f, -0.9113f, 0.1075f, -0.1745f, 0.9653f, -0.1517f, 1.5508f,
                         0.4134f, -0.0715f, 0.5517f, -0.3632f, -0.1922f, -0.9497f, 0.2503f, -0.2921f};
const float kStd[16] = {2.8184f, 1.4541f, 2.3275f, 2.6558f, 1.2196f, 1.7708f, 2.6052f, 2.0743f,
                        3.2687f, 2.1526f, 2.8652f, 1.5579f, 1.6382f, 1.1253f, 2.8251f, 1.9160f};

struct ConvW {               // one conv layer: user weight (borrowed) + packed copy (owned)
    std::vector<int64_t> shape;   // [Cout, Cin, kt, kh, kw] or [Cout, Cin, kh, kw]
    const float* w_user = nullptr;
    const float* b_user = nullptr;
    float* packed = nullptr;
    bf16* packed3 = nullptr;          // three bf16 planes of the packed weights (see conv_igemm_x3_kernel)
    unsigned short* packed2h = nullptr;   // two fp16 planes of the row-scaled packed weights (conv_igemm_x3_kernel<true>)
    float* w2_scale = nullptr;        // [2][Cout]: scale | 1 / scale
    bf16* packed3_up = nullptr;       // upsample convolutions: the four 2x2 phase kernels, three bf16 planes each (launch_conv_up_phases)
    bool upsample = false;            // a 3x3 spatial convolution read through a nearest x2 upsample (Resample upsample2d/3d)
    int Cout = 0, Cin = 0, kt = 1, kh = 1, kw = 1, ldw = 0, ldw3 = 0;
};
struct Tens {
    float* p = nullptr; int T = 0, H = 0, W = 0, C = 0;
    float bound = 0.f;                // > 0: every |element| <= bound, guaranteed by the producer (RMS_norm [+ SiLU]); 0: unknown
    float plane_scale = 0.f;          // > 0: the slot holds two fp16 planes [pixel][C] (hi at p, lo behind it) of value * plane_scale instead of fp32
    long elems() const { return (long)T * H * W * C; }
};

}  // namespace

struct svi_vae {
    int device = -1;                                    // claimed by the first call that touches the GPU (svi_claim_device)
    std::map<std::string, ConvW> convs;                 // key = layer prefix without ".weight"
    std::map<std::string, const float*> gammas;         // key = full name
    std::map<std::string, std::vector<int64_t>> gamma_shapes;
    std::map<std::string, float> gamma_bound;           // key = full name: sqrt(C) * max|gamma| (0: unusable, see svi_vae_bind_weight)
    // workspace
    char* pool = nullptr;
    size_t pool_bytes = 0, slot_bytes = 0;
    int nslots = 0;
    std::vector<int> slot_used;
    float* consts = nullptr;                            // mean[16] | inv_std[16]
    float* attn_scratch = nullptr; size_t attn_bytes = 0;
    bool dry = false;
    long dry_max = 0;
    bool pack_pending = false;                          // weight packing kernels enqueued on the null stream since the last call
    float* blend_w = nullptr; size_t blend_bytes = 0;   // mask accumulator of the tiled paths ([H, W] plane)
    const TileWin* win = nullptr;                       // set while a tile of svi_vae_tiled_* runs through the graphs
    float* blend_values = nullptr;
};

namespace {

void add_conv(svi_vae* h, const std::string& name, int co, int ci, int kt, int kh, int kw, bool is2d = false) {
    ConvW c;
    c.Cout = co; c.Cin = ci; c.kt = kt; c.kh = kh; c.kw = kw; c.ldw = (ci + 3) / 4 * 4;
    if (is2d) c.shape = {co, ci, kh, kw};
    else c.shape = {co, ci, kt, kh, kw};
    h->convs[name] = c;
}
void add_gamma(svi_vae* h, const std::string& name, int c, bool images) {
    h->gammas[name] = nullptr;
    if (images) h->gamma_shapes[name] = {c, 1, 1};
    else h->gamma_shapes[name] = {c, 1, 1, 1};
}
void add_res(svi_vae* h, const std::string& p, int ci, int co) {
    add_gamma(h, p + "residual.0.gamma", ci, false);
    add_conv(h, p + "residual.2", co, ci, 3, 3, 3);
    add_gamma(h, p + "residual.3.gamma", co, false);
    add_conv(h, p + "residual.6", co, co, 3, 3, 3);
    if (ci != co) add_conv(h, p + "shortcut", co, ci, 1, 1, 1);
}
void add_attn(svi_vae* h, const std::string& p, int c) {
    add_gamma(h, p + "norm.gamma", c, true);
    add_conv(h, p + "to_qkv", 3 * c, c, 1, 1, 1, true);
    add_conv(h, p + "proj", c, c, 1, 1, 1, true);
}

// architecture table: dim 96, z 16, mult (1,2,4,4), 2 res blocks, temporal down (F,T,T)  (vae:494-517)
void declare_architecture(svi_vae* h) {
    const int base = 96, z = 16, mult[4] = {1, 2, 4, 4};
    const bool tdown[3] = {false, true, true};
    const std::string e = "model.encoder.", d = "model.decoder.";
    int dims[5] = {base, base * mult[0], base * mult[1], base * mult[2], base * mult[3]};
    add_conv(h, e + "conv1", dims[0], 3, 3, 3, 3);
    int idx = 0;
    for (int i = 0; i < 4; ++i) {
        int di = dims[i];
        const int dout = dims[i + 1];
        for (int r = 0; r < 2; ++r) { add_res(h, e + "downsamples." + std::to_string(idx++) + ".", di, dout); di = dout; }
        if (i != 3) {
            const std::string p = e + "downsamples." + std::to_string(idx++) + ".";
            add_conv(h, p + "resample.1", dout, dout, 1, 3, 3, true);
            if (tdown[i]) add_conv(h, p + "time_conv", dout, dout, 3, 1, 1);
        }
    }
    const int top = dims[4];
    add_res(h, e + "middle.0.", top, top); add_attn(h, e + "middle.1.", top); add_res(h, e + "middle.2.", top, top);
    add_gamma(h, e + "head.0.gamma", top, false);
    add_conv(h, e + "head.2", 2 * z, top, 3, 3, 3);
    add_conv(h, "model.conv1", 2 * z, 2 * z, 1, 1, 1);
    add_conv(h, "model.conv2", z, z, 1, 1, 1);
    int dd[5] = {base * mult[3], base * mult[3], base * mult[2], base * mult[1], base * mult[0]};
    add_conv(h, d + "conv1", dd[0], z, 3, 3, 3);
    add_res(h, d + "middle.0.", dd[0], dd[0]); add_attn(h, d + "middle.1.", dd[0]); add_res(h, d + "middle.2.", dd[0], dd[0]);
    const bool tup[3] = {true, true, false};
    idx = 0;
    for (int i = 0; i < 4; ++i) {
        int di = dd[i];
        const int dout = dd[i + 1];
        if (i >= 1) di = di / 2;
        for (int r = 0; r < 3; ++r) { add_res(h, d + "upsamples." + std::to_string(idx++) + ".", di, dout); di = dout; }
        if (i != 3) {
            const std::string p = d + "upsamples." + std::to_string(idx++) + ".";
            add_conv(h, p + "resample.1", dout / 2, dout, 1, 3, 3, true);
            h->convs[p + "resample.1"].upsample = true;
            if (tup[i]) add_conv(h, p + "time_conv", dout * 2, dout, 3, 1, 1);
        }
    }
    add_gamma(h, d + "head.0.gamma", dd[4], false);
    add_conv(h, d + "head.2", 3, dd[4], 3, 3, 3);
}

// ---- tensor pool: fixed number of equally sized slots, sized by a dry run of the same code path -----------------------
Tens alloc_t(svi_vae* h, int T, int H, int W, int C) {
    Tens t; t.T = T; t.H = H; t.W = W; t.C = C;
    if (h->dry) { if (t.elems() > h->dry_max) h->dry_max = t.elems(); t.p = nullptr; return t; }
    for (int i = 0; i < h->nslots; ++i)
        if (!h->slot_used[i]) { h->slot_used[i] = 1; t.p = reinterpret_cast<float*>(h->pool + (size_t)i * h->slot_bytes); return t; }
    t.p = nullptr;       // caller checks
    return t;
}
void free_t(svi_vae* h, Tens& t) {
    if (h->dry || !t.p) { t.p = nullptr; return; }
    const size_t i = (reinterpret_cast<char*>(t.p) - h->pool) / h->slot_bytes;
    h->slot_used[i] = 0;
    t.p = nullptr;
}
#define NEED(t) do { if (!h->dry && !(t).p) { svi_set_error("VAE tensor pool exhausted"); return SVI_ERR_OOM; } } while (0)

static float scale_of_bound(float bound) {          // largest power of two s with bound * s <= 2^15 (0: no usable bound)
    if (!(bound > 0.f && bound < 1e30f)) return 0.f;
    int e = 0;
    (void)frexpf(bound, &e);
    return ldexpf(1.0f, 15 - e);
}

// The parts of a convolution's launch parameters that do not depend on where tensors live: geometry, weights, input scale.
static ConvP conv_params(const ConvW& c, const Tens& in, int stride_t, bool causal_pad, int sh, int ups, int zero_frame0, bool down_pad, int* To_, int* Ho_, int* Wo_) {
    ConvP p{};
    p.Ti = in.T; p.Hi = in.H; p.Wi = in.W; p.Cin = (c.Cin + 3) / 4 * 4; p.ld_in = in.C;
    p.kt = c.kt; p.kh = c.kh; p.kw = c.kw; p.st = stride_t; p.sh = sh; p.sw = sh;
    p.pt = causal_pad ? c.kt - 1 : 0; p.ph = down_pad ? 0 : c.kh / 2; p.pw = down_pad ? 0 : c.kw / 2;
    p.ups = ups; p.zero_frame0 = zero_frame0;
    const int Hv = ups ? 2 * in.H : in.H, Wv = ups ? 2 * in.W : in.W;
    *To_ = causal_pad ? in.T : (in.T - c.kt) / stride_t + 1;
    *Ho_ = down_pad ? Hv / 2 : Hv; *Wo_ = down_pad ? Wv / 2 : Wv;
    p.w = c.packed; p.ld_w = c.ldw; p.bias = c.b_user;
    p.w3 = c.packed3; p.ld_w3 = c.ldw3; p.plane_w3 = (long)c.kt * c.kh * c.kw * c.Cout * c.ldw3;
    p.w2h = c.packed2h; p.w2_inv = c.w2_scale ? c.w2_scale + c.Cout : nullptr;
    p.in_scale = scale_of_bound(in.bound);
    p.To = *To_; p.Ho = *Ho_; p.Wo = *Wo_; p.Cout = c.Cout;
    return p;
}

// Would `name`, fed by RMS_norm `gname` of a tensor shaped like `x`, take its input as fp16 planes?  (stride 1, causal, same-size: the residual-block convolutions)
static float planes_scale_for(svi_vae* h, const std::string& name, const std::string& gname, const Tens& x) {
    if (h->dry) return 0.f;
    auto gb = h->gamma_bound.find(gname);
    Tens like = x;
    like.bound = gb == h->gamma_bound.end() ? 0.f : gb->second;
    int To, Ho, Wo;
    ConvP p = conv_params(h->convs.at(name), like, 1, true, 1, 0, 0, false, &To, &Ho, &Wo);
    return conv_planes_ok(p) ? p.in_scale : 0.f;
}

svi_status conv_layer(svi_vae* h, const std::string& name, const Tens& in, Tens* out, hipStream_t st, int stride_t = 1,
                      bool causal_pad = true, int sh = 1, int ups = 0, int zero_frame0 = 0, const Tens* res = nullptr,
                      bool down_pad = false) {
    const ConvW& c = h->convs.at(name);
    int To, Ho, Wo;
    ConvP p = conv_params(c, in, stride_t, causal_pad, sh, ups, zero_frame0, down_pad, &To, &Ho, &Wo);
    *out = alloc_t(h, To, Ho, Wo, c.Cout < 4 ? 4 : c.Cout);
    NEED(*out);
    if (h->dry) return SVI_OK;
    p.in = in.p; p.out = out->p; p.ld_out = out->C;
    p.res = res ? res->p : nullptr; p.ld_res = res ? res->C : 0;
    SviProfScope _p(PROF_VAE_CONV, st);
    if (in.plane_scale > 0.f) {          // the producer wrote fp16 planes for this layer (planes_scale_for said it would take them)
        SVI_REQUIRE(in.plane_scale == p.in_scale, "conv %s: input planes carry scale %g, the layer expects %g", name.c_str(), in.plane_scale, p.in_scale);
        p.in = nullptr;
        p.in_h = reinterpret_cast<const unsigned short*>(in.p);
        p.in_l = p.in_h + in.elems();
        return launch_conv_planes(p, st);
    }
    if (ups && c.packed3_up && svi_switches().vae_up_phases && !res) {
        ConvP q = p;                     // one phase's launch parameters, to ask whether the three-term kernel takes it
        q.ups = 0; q.kh = q.kw = 2; q.Ho = q.Hi; q.Wo = q.Wi; q.plane_w3 = (long)q.kt * 4 * q.Cout * q.ld_w3;
        if (conv_x3_ok(q)) return launch_conv_up_phases(p, c.packed3_up, st);
    }
    return launch_conv(p, st);
}

svi_status norm_act(svi_vae* h, const std::string& gname, const Tens& in, Tens* out, int do_silu, hipStream_t st, float plane_scale = 0.f) {
    *out = alloc_t(h, in.T, in.H, in.W, in.C);
    NEED(*out);
    if (h->dry) return SVI_OK;
    {   // |x_c| / max(||x||, eps) <= 1, so |out_c| <= sqrt(C) |gamma_c|; SiLU never increases a magnitude
        auto gb = h->gamma_bound.find(gname);
        out->bound = gb == h->gamma_bound.end() ? 0.f : gb->second;
    }
    SviProfScope _p(PROF_VAE_OTHER, st);
    if (plane_scale > 0.f) {             // the only consumer is a convolution that multiplies fp16 word pairs: write those instead of fp32
        out->plane_scale = plane_scale;
        unsigned short* ph = reinterpret_cast<unsigned short*>(out->p);
        return launch_rms_silu_planes(in.p, ph, ph + in.elems(), (long)in.T * in.H * in.W, in.C, h->gammas.at(gname), do_silu, plane_scale, st);
    }
    return launch_rms_silu(in.p, out->p, (long)in.T * in.H * in.W, in.C, h->gammas.at(gname), do_silu, st);
}

// ResidualBlock (vae:198-232): x <- conv2(silu(norm(conv1(silu(norm(x)))))) + shortcut(x); consumes x.
svi_status res_block(svi_vae* h, const std::string& p, Tens* x, hipStream_t st) {
    Tens n, hmid, out, skip;
    const bool has_sc = h->convs.count(p + "shortcut") > 0;
    if (has_sc) SVI_TRY(conv_layer(h, p + "shortcut", *x, &skip, st));
    SVI_TRY(norm_act(h, p + "residual.0.gamma", *x, &n, 1, st, planes_scale_for(h, p + "residual.2", p + "residual.0.gamma", *x)));
    SVI_TRY(conv_layer(h, p + "residual.2", n, &hmid, st));
    free_t(h, n);
    SVI_TRY(norm_act(h, p + "residual.3.gamma", hmid, &n, 1, st, planes_scale_for(h, p + "residual.6", p + "residual.3.gamma", hmid)));
    free_t(h, hmid);
    SVI_TRY(conv_layer(h, p + "residual.6", n, &out, st, 1, true, 1, 0, 0, has_sc ? &skip : x));
    free_t(h, n);
    if (has_sc) free_t(h, skip);
    free_t(h, *x);
    *x = out;
    return SVI_OK;
}

// AttentionBlock (vae:235-273): per frame, single head over h*w tokens with head dim C.
svi_status attn_block(svi_vae* h, const std::string& p, Tens* x, hipStream_t st) {
    Tens n, qkv, out;
    SVI_TRY(norm_act(h, p + "norm.gamma", *x, &n, 0, st));
    SVI_TRY(conv_layer(h, p + "to_qkv", n, &qkv, st));
    free_t(h, n);
    const int C = x->C, hw = x->H * x->W;
    const int ldp = (hw + 3) / 4 * 4;
    Tens att = alloc_t(h, x->T, x->H, x->W, C);
    NEED(att);
    if (!h->dry) {
        const size_t need = ((size_t)hw * ldp + (size_t)C * ldp) * 4;
        if (h->attn_bytes < need) {
            if (h->attn_scratch) SVI_CHECK_HIP(hipFree(h->attn_scratch));
            h->attn_scratch = nullptr; h->attn_bytes = 0;
            hipError_t e = hipMalloc((void**)&h->attn_scratch, need);
            if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B VAE attention scratch) failed", need); return SVI_ERR_OOM; }
            h->attn_bytes = need;
        }
        // pad columns (hw..ldp-1) of S and V^T take part in the P·V contraction: they must read as zeros, and the
        // scratch may hold data from an earlier call with another hw
        if (ldp != hw) SVI_CHECK_HIP(hipMemsetAsync(h->attn_scratch, 0, need, st));
        float* S = h->attn_scratch;                 // [hw, ldp]
        float* VT = S + (size_t)hw * ldp;           // [C, ldp]
        SviProfScope _pp(PROF_VAE_OTHER, st);
        for (int t = 0; t < x->T; ++t) {
            const float* q = qkv.p + (long)t * hw * 3 * C;
            ConvP g{};                              // S = Q K^T   (1x1 "conv": pixels = queries, Cout = keys)
            g.in = q; g.Ti = 1; g.Hi = 1; g.Wi = hw; g.Cin = C; g.ld_in = 3 * C;
            g.w = q + C; g.ld_w = 3 * C; g.bias = nullptr;
            g.out = S; g.To = 1; g.Ho = 1; g.Wo = hw; g.Cout = hw; g.ld_out = ldp;
            g.kt = g.kh = g.kw = 1; g.st = g.sh = g.sw = 1;
            SVI_TRY(launch_conv(g, st));
            hipLaunchKernelGGL(softmax_rows_kernel, dim3((hw + 3) / 4), dim3(256), 0, st, S, hw, hw, ldp, 1.0f / sqrtf((float)C));
            hipLaunchKernelGGL(transpose_f32_kernel, dim3((C + 63) / 64, (hw + 63) / 64), dim3(256), 0, st, q + 2 * C, 3 * C, VT, ldp, hw, C);
            SVI_LAUNCH_CHECK();
            ConvP o{};                              // O = P V     (Cin = keys, weights = V^T)
            o.in = S; o.Ti = 1; o.Hi = 1; o.Wi = hw; o.Cin = ldp; o.ld_in = ldp;
            o.w = VT; o.ld_w = ldp; o.bias = nullptr;
            o.out = att.p + (long)t * hw * C; o.To = 1; o.Ho = 1; o.Wo = hw; o.Cout = C; o.ld_out = C;
            o.kt = o.kh = o.kw = 1; o.st = o.sh = o.sw = 1;
            SVI_TRY(launch_conv(o, st));
        }
    }
    free_t(h, qkv);
    SVI_TRY(conv_layer(h, p + "proj", att, &out, st, 1, true, 1, 0, 0, x));
    free_t(h, att);
    free_t(h, *x);
    *x = out;
    return SVI_OK;
}

// Resample 'upsample2d' / 'upsample3d' (vae:120-160)
svi_status upsample_block(svi_vae* h, const std::string& p, Tens* x, bool temporal, hipStream_t st) {
    if (temporal && x->T > 1) {
        const ConvW& c = h->convs.at(p + "time_conv");
        Tens up = alloc_t(h, 1 + 2 * (x->T - 1), x->H, x->W, x->C);
        NEED(up);
        if (!h->dry) {
            ConvP q{};
            q.in = x->p; q.Ti = x->T; q.Hi = x->H; q.Wi = x->W; q.Cin = c.Cin; q.ld_in = x->C;
            q.w = c.packed; q.ld_w = c.ldw; q.bias = c.b_user;
            q.w3 = c.packed3; q.ld_w3 = c.ldw3; q.plane_w3 = (long)c.kt * c.kh * c.kw * c.Cout * c.ldw3;
            q.out = up.p; q.To = x->T; q.Ho = x->H; q.Wo = x->W; q.Cout = c.Cout; q.ld_out = up.C;
            q.kt = 3; q.kh = q.kw = 1; q.st = q.sh = q.sw = 1; q.pt = 2;
            q.zero_frame0 = 1; q.t_begin = 1; q.out_mode = 1;
            { SviProfScope _p(PROF_VAE_CONV, st); SVI_TRY(launch_conv(q, st)); }
            SVI_CHECK_HIP(hipMemcpyAsync(up.p, x->p, (size_t)x->H * x->W * x->C * 4, hipMemcpyDeviceToDevice, st));
        }
        free_t(h, *x);
        *x = up;
    }
    Tens out;
    SVI_TRY(conv_layer(h, p + "resample.1", *x, &out, st, 1, true, 1, /*ups=*/1));
    free_t(h, *x);
    *x = out;
    return SVI_OK;
}

// Resample 'downsample2d' / 'downsample3d' (vae:157-173)
svi_status downsample_block(svi_vae* h, const std::string& p, Tens* x, bool temporal, hipStream_t st) {
    Tens sp;
    SVI_TRY(conv_layer(h, p + "resample.1", *x, &sp, st, 1, true, /*sh=*/2, 0, 0, nullptr, /*down_pad=*/true));
    free_t(h, *x);
    *x = sp;
    if (temporal && x->T > 1) {
        const ConvW& c = h->convs.at(p + "time_conv");
        const int To = (x->T - 3) / 2 + 1;
        Tens dn = alloc_t(h, 1 + To, x->H, x->W, x->C);
        NEED(dn);
        if (!h->dry) {
            ConvP q{};
            q.in = x->p; q.Ti = x->T; q.Hi = x->H; q.Wi = x->W; q.Cin = c.Cin; q.ld_in = x->C;
            q.w = c.packed; q.ld_w = c.ldw; q.bias = c.b_user;
            q.w3 = c.packed3; q.ld_w3 = c.ldw3; q.plane_w3 = (long)c.kt * c.kh * c.kw * c.Cout * c.ldw3;
            q.out = dn.p; q.To = To; q.Ho = x->H; q.Wo = x->W; q.Cout = c.Cout; q.ld_out = dn.C;
            q.kt = 3; q.kh = q.kw = 1; q.st = 2; q.sh = q.sw = 1; q.pt = 0;
            q.t_out_off = 1;
            { SviProfScope _p(PROF_VAE_CONV, st); SVI_TRY(launch_conv(q, st)); }
            SVI_CHECK_HIP(hipMemcpyAsync(dn.p, x->p, (size_t)x->H * x->W * x->C * 4, hipMemcpyDeviceToDevice, st));
        }
        free_t(h, *x);
        *x = dn;
    }
    return SVI_OK;
}

svi_status decode_graph(svi_vae* h, const float* latents, float* video, int T, int hh, int ww, hipStream_t st) {
    const std::string d = "model.decoder.";
    Tens z = alloc_t(h, T, hh, ww, 16);
    NEED(z);
    if (!h->dry) {
        const long n = (long)T * hh * ww * 16;
        if (h->win) hipLaunchKernelGGL(latent_in_win_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, latents, z.p, T, *h->win, h->consts, h->consts + 16);
        else hipLaunchKernelGGL(latent_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, latents, z.p, (long)T * hh * ww, h->consts, h->consts + 16);
        SVI_LAUNCH_CHECK();
    }
    Tens x, t2;
    SVI_TRY(conv_layer(h, "model.conv2", z, &t2, st));
    free_t(h, z);
    SVI_TRY(conv_layer(h, d + "conv1", t2, &x, st));
    free_t(h, t2);
    SVI_TRY(res_block(h, d + "middle.0.", &x, st));
    SVI_TRY(attn_block(h, d + "middle.1.", &x, st));
    SVI_TRY(res_block(h, d + "middle.2.", &x, st));
    const bool tup[3] = {true, true, false};
    int idx = 0;
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 3; ++r) SVI_TRY(res_block(h, d + "upsamples." + std::to_string(idx++) + ".", &x, st));
        if (i != 3) SVI_TRY(upsample_block(h, d + "upsamples." + std::to_string(idx++) + ".", &x, tup[i], st));
    }
    Tens n, rgb;
    SVI_TRY(norm_act(h, d + "head.0.gamma", x, &n, 1, st, planes_scale_for(h, d + "head.2", d + "head.0.gamma", x)));
    free_t(h, x);
    SVI_TRY(conv_layer(h, d + "head.2", n, &rgb, st));
    free_t(h, n);
    if (!h->dry) {
        const long thw = (long)rgb.T * rgb.H * rgb.W;
        if (h->win) hipLaunchKernelGGL(tile_blend_kernel, dim3((unsigned)((thw + 255) / 256)), dim3(256), 0, st, rgb.p, rgb.C, video, h->blend_w, 3, rgb.T,
                                       rgb.H, rgb.W, *h->win, 0, h->consts, h->consts + 16);
        else hipLaunchKernelGGL(video_out_kernel, dim3((unsigned)((thw * 3 + 255) / 256)), dim3(256), 0, st, rgb.p, rgb.C, video, thw);
        SVI_LAUNCH_CHECK();
    }
    free_t(h, rgb);
    return SVI_OK;
}

svi_status encode_graph(svi_vae* h, const float* video, float* latents, int T, int H, int W, hipStream_t st) {
    const std::string e = "model.encoder.";
    Tens v = alloc_t(h, T, H, W, 4);
    NEED(v);
    if (!h->dry) {
        const long thw = (long)T * H * W;
        if (h->win) hipLaunchKernelGGL(video_in_win_kernel, dim3((unsigned)((thw * 4 + 255) / 256)), dim3(256), 0, st, video, v.p, T, *h->win);
        else hipLaunchKernelGGL(video_in_kernel, dim3((unsigned)((thw * 4 + 255) / 256)), dim3(256), 0, st, video, v.p, thw);
        SVI_LAUNCH_CHECK();
    }
    Tens x;
    SVI_TRY(conv_layer(h, e + "conv1", v, &x, st));
    free_t(h, v);
    const bool tdown[3] = {false, true, true};
    int idx = 0;
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 2; ++r) SVI_TRY(res_block(h, e + "downsamples." + std::to_string(idx++) + ".", &x, st));
        if (i != 3) SVI_TRY(downsample_block(h, e + "downsamples." + std::to_string(idx++) + ".", &x, tdown[i], st));
    }
    SVI_TRY(res_block(h, e + "middle.0.", &x, st));
    SVI_TRY(attn_block(h, e + "middle.1.", &x, st));
    SVI_TRY(res_block(h, e + "middle.2.", &x, st));
    Tens n, hd, mu;
    SVI_TRY(norm_act(h, e + "head.0.gamma", x, &n, 1, st, planes_scale_for(h, e + "head.2", e + "head.0.gamma", x)));
    free_t(h, x);
    SVI_TRY(conv_layer(h, e + "head.2", n, &hd, st));
    free_t(h, n);
    SVI_TRY(conv_layer(h, "model.conv1", hd, &mu, st));
    free_t(h, hd);
    if (!h->dry) {
        const long thw = (long)mu.T * mu.H * mu.W;
        if (h->win) hipLaunchKernelGGL(tile_blend_kernel, dim3((unsigned)((thw + 255) / 256)), dim3(256), 0, st, mu.p, mu.C, latents, h->blend_w, 16, mu.T,
                                       mu.H, mu.W, *h->win, 1, h->consts, h->consts + 16);
        else hipLaunchKernelGGL(latent_out_kernel, dim3((unsigned)((thw * 16 + 255) / 256)), dim3(256), 0, st, mu.p, mu.C, latents, thw, h->consts, h->consts + 16);
        SVI_LAUNCH_CHECK();
    }
    free_t(h, mu);
    return SVI_OK;
}

svi_status ensure_pool(svi_vae* h, long max_elems) {
    const size_t slot = ((size_t)max_elems * 4 + 255) & ~(size_t)255;
    const int nslots = 5;
    if (h->pool && h->slot_bytes >= slot) return SVI_OK;
    if (h->pool) { SVI_CHECK_HIP(hipFree(h->pool)); h->pool = nullptr; }
    hipError_t e = hipMalloc((void**)&h->pool, slot * nslots);
    if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B VAE activation pool) failed: %s", slot * nslots, hipGetErrorString(e)); return SVI_ERR_OOM; }
    h->slot_bytes = slot; h->nslots = nslots; h->pool_bytes = slot * nslots;
    h->slot_used.assign(nslots, 0);
    return SVI_OK;
}

svi_status check_bound(svi_vae* h) {
    for (auto& kv : h->convs)
        if (!kv.second.w_user || !kv.second.b_user) { svi_set_error("VAE parameter '%s.weight/.bias' was never bound", kv.first.c_str()); return SVI_ERR_UNBOUND; }
    for (auto& kv : h->gammas)
        if (!kv.second) { svi_set_error("VAE parameter '%s' was never bound", kv.first.c_str()); return SVI_ERR_UNBOUND; }
    return SVI_OK;
}

}  // namespace

extern "C" svi_status svi_vae_create(svi_vae** out) {
    SVI_REQUIRE(out, "svi_vae_create: null argument");
    svi_vae* h = new (std::nothrow) svi_vae();
    if (!h) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    declare_architecture(h);
    *out = h;
    return SVI_OK;
}

extern "C" svi_status svi_vae_destroy(svi_vae* h) {
    if (!h) return SVI_OK;
    for (auto& kv : h->convs) {
        if (kv.second.packed) (void)hipFree(kv.second.packed);
        if (kv.second.packed3) (void)hipFree(kv.second.packed3);
        if (kv.second.packed2h) (void)hipFree(kv.second.packed2h);
        if (kv.second.packed3_up) (void)hipFree(kv.second.packed3_up);
        if (kv.second.w2_scale) (void)hipFree(kv.second.w2_scale);
    }
    if (h->pool) (void)hipFree(h->pool);
    if (h->consts) (void)hipFree(h->consts);
    if (h->attn_scratch) (void)hipFree(h->attn_scratch);
    if (h->blend_w) (void)hipFree(h->blend_w);
    delete h;
    return SVI_OK;
}

extern "C" svi_status svi_vae_bind_weight(svi_vae* h, const char* name, const void* dev_ptr, svi_dtype dtype,
                                          const int64_t* shape, int32_t rank) {
    SVI_REQUIRE(h && name && dev_ptr && shape, "svi_vae_bind_weight: null argument");
    SVI_REQUIRE(dtype == SVI_F32, "VAE parameter '%s' must be fp32", name);
    SVI_REQUIRE(((uintptr_t)dev_ptr % 16) == 0, "VAE parameter '%s' is not 16-byte aligned", name);
    SVI_REQUIRE_DEVICE(h);
    const std::string key(name);
    auto shape_ok = [&](const std::vector<int64_t>& want) {
        if ((int)want.size() != rank) return false;
        for (int i = 0; i < rank; ++i) if (want[i] != shape[i]) return false;
        return true;
    };
    auto g = h->gammas.find(key);
    if (g != h->gammas.end()) {
        if (!shape_ok(h->gamma_shapes[key])) { svi_set_error("shape mismatch for '%s'", name); return SVI_ERR_INVALID; }
        g->second = reinterpret_cast<const float*>(dev_ptr);
        {   // the static bound the fp16 two-term convolution relies on; a gain vector with a dead or wildly smaller channel (its
            // activations would sit in fp16's subnormals after the common scale) keeps the three-term kernel for its consumers
            std::vector<float> host((size_t)shape[0]);
            SVI_CHECK_HIP(hipMemcpy(host.data(), dev_ptr, host.size() * 4, hipMemcpyDeviceToHost));
            float mx = 0.f, mn = INFINITY;
            for (float v : host) { mx = fmaxf(mx, fabsf(v)); mn = fminf(mn, fabsf(v)); }
            const bool ok = mx > 0.f && mx < 1e30f && mn >= mx * (1.0f / 256.0f);
            h->gamma_bound[key] = ok ? sqrtf((float)shape[0]) * mx * 1.0001f : 0.f;
        }
        return SVI_OK;
    }
    const size_t dot = key.rfind('.');
    if (dot == std::string::npos) { svi_set_error("unknown VAE parameter '%s'", name); return SVI_ERR_INVALID; }
    const std::string layer = key.substr(0, dot), leaf = key.substr(dot + 1);
    auto c = h->convs.find(layer);
    if (c == h->convs.end() || (leaf != "weight" && leaf != "bias")) { svi_set_error("unknown VAE parameter '%s'", name); return SVI_ERR_INVALID; }
    ConvW& cw = c->second;
    if (leaf == "bias") {
        if (!(rank == 1 && shape[0] == cw.Cout)) { svi_set_error("shape mismatch for '%s'", name); return SVI_ERR_INVALID; }
        cw.b_user = reinterpret_cast<const float*>(dev_ptr);
        return SVI_OK;
    }
    if (!shape_ok(cw.shape)) { svi_set_error("shape mismatch for '%s'", name); return SVI_ERR_INVALID; }
    cw.w_user = reinterpret_cast<const float*>(dev_ptr);
    const int taps = cw.kt * cw.kh * cw.kw;
    const size_t n = (size_t)taps * cw.Cout * cw.ldw;
    if (!cw.packed) {
        hipError_t e = hipMalloc((void**)&cw.packed, n * 4);
        if (e != hipSuccess) { svi_set_error("hipMalloc(packed VAE weight) failed: %s", hipGetErrorString(e)); return SVI_ERR_OOM; }
    }
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, cw.w_user, cw.packed, cw.Cout, cw.Cin, taps, cw.ldw);
    const bool narrow_head = cw.Cout <= 32 && cw.kh == 3 && cw.kw == 3 && cw.Cin % 32 == 0;       // conv_dma2h_kernel<1> (two-term fp16 form on the producer's planes)
    if (cw.Cout >= 64 || narrow_head) {               // layers wide enough for conv_igemm_x3_kernel get the three-term bf16 form and the two-term fp16 form
        cw.ldw3 = (cw.Cin + 31) / 32 * 32;
        const size_t n3 = (size_t)taps * cw.Cout * cw.ldw3;
        if (!cw.packed3) {
            hipError_t e3 = hipMalloc((void**)&cw.packed3, 3 * n3 * 2);
            if (e3 != hipSuccess) { svi_set_error("hipMalloc(split VAE weight) failed: %s", hipGetErrorString(e3)); return SVI_ERR_OOM; }
        }
        hipLaunchKernelGGL(pack_weight_x3_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, 0, cw.w_user, cw.packed3, cw.Cout, cw.Cin, taps, cw.ldw3);
        if (!cw.packed2h) {
            hipError_t e2 = hipMalloc((void**)&cw.packed2h, 2 * n3 * 2);
            if (e2 == hipSuccess) e2 = hipMalloc((void**)&cw.w2_scale, (size_t)2 * cw.Cout * 4);
            if (e2 != hipSuccess) { svi_set_error("hipMalloc(fp16 split VAE weight) failed: %s", hipGetErrorString(e2)); return SVI_ERR_OOM; }
        }
        if (cw.upsample && cw.kt == 1 && cw.kh == 3 && cw.kw == 3) {
            const size_t nu = (size_t)4 * 3 * 4 * cw.Cout * cw.ldw3;
            if (!cw.packed3_up) {
                hipError_t eu = hipMalloc((void**)&cw.packed3_up, nu * 2);
                if (eu != hipSuccess) { svi_set_error("hipMalloc(upsample phase weights) failed: %s", hipGetErrorString(eu)); return SVI_ERR_OOM; }
            }
            const size_t nt = (size_t)4 * 4 * cw.Cout * cw.ldw3;
            hipLaunchKernelGGL(pack_weight_up_x3_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, cw.w_user, cw.packed3_up, cw.Cout, cw.Cin, cw.ldw3);
        }
        hipLaunchKernelGGL(weight_row_scale_kernel, dim3(cw.Cout), dim3(256), 0, 0, cw.w_user, cw.w2_scale, cw.w2_scale + cw.Cout, (long)cw.Cin * taps);
        hipLaunchKernelGGL(pack_weight_x2h_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, 0, cw.w_user, cw.w2_scale, cw.packed2h, cw.Cout, cw.Cin, taps, cw.ldw3);
    }
    SVI_LAUNCH_CHECK();
    h->pack_pending = true;
    return SVI_OK;
}

extern "C" svi_status svi_vae_check_bound(svi_vae* h) {
    SVI_REQUIRE(h, "null handle");
    return check_bound(h);
}

static svi_status vae_prepare(svi_vae* h) {
    SVI_TRY(check_bound(h));
    if (h->pack_pending) {       // packing ran on the null stream; the caller's stream need not be ordered after it
        SVI_CHECK_HIP(hipStreamSynchronize(nullptr));
        h->pack_pending = false;
    }
    if (!h->consts) {
        float host[32];
        for (int i = 0; i < 16; ++i) { host[i] = kMean[i]; host[16 + i] = 1.0f / kStd[i]; }
        SVI_CHECK_HIP(hipMalloc((void**)&h->consts, sizeof(host)));
        SVI_CHECK_HIP(hipMemcpy(h->consts, host, sizeof(host), hipMemcpyHostToDevice));
        SVI_CHECK_HIP(hipDeviceSynchronize());
    }
    return SVI_OK;
}

extern "C" svi_status svi_vae_decode(svi_vae* h, const float* latents, float* video, int32_t T, int32_t hh, int32_t ww,
                                     svi_stream stream) {
    SVI_REQUIRE(h && latents && video && T > 0 && hh > 0 && ww > 0, "svi_vae_decode: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_TRY(vae_prepare(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    h->dry = true; h->dry_max = 0;
    SVI_TRY(decode_graph(h, latents, video, T, hh, ww, st));
    h->dry = false;
    SVI_TRY(ensure_pool(h, h->dry_max));
    h->slot_used.assign(h->nslots, 0);
    return decode_graph(h, latents, video, T, hh, ww, st);
}

extern "C" svi_status svi_vae_encode(svi_vae* h, const float* video, float* latents, int32_t T, int32_t H, int32_t W,
                                     svi_stream stream) {
    SVI_REQUIRE(h && video && latents && T > 0 && H > 0 && W > 0, "svi_vae_encode: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE((T - 1) % 4 == 0 && H % 8 == 0 && W % 8 == 0, "svi_vae_encode: needs T = 1+4k frames and H, W multiples of 8 (got %d, %d, %d)", T, H, W);
    SVI_TRY(vae_prepare(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    h->dry = true; h->dry_max = 0;
    SVI_TRY(encode_graph(h, video, latents, T, H, W, st));
    h->dry = false;
    SVI_TRY(ensure_pool(h, h->dry_max));
    h->slot_used.assign(h->nslots, 0);
    return encode_graph(h, video, latents, T, H, W, st);
}

// ---- tiled paths ------------------------------------------------------------------------------------------------------------
namespace {
struct WinScope {            // h->win points at a stack object while one tile runs through a graph: never leave it dangling on an error return
    svi_vae* h;
    WinScope(svi_vae* hh, const TileWin* w) : h(hh) { h->win = w; }
    ~WinScope() { h->win = nullptr; }
};
// The task list of vae:648-655 / :697-704: tiles start every `stride`; a start is dropped when the previous tile already reaches
// the far edge.
std::vector<std::pair<int, int>> tile_starts(int full, int size, int stride) {
    std::vector<std::pair<int, int>> t;
    for (int a = 0; a < full; a += stride) {
        if (a - stride >= 0 && a - stride + size >= full) continue;
        t.push_back({a, a + size});
    }
    return t;
}
svi_status ensure_blend(svi_vae* h, size_t bytes) {
    if (h->blend_bytes >= bytes) return SVI_OK;
    if (h->blend_w) { SVI_CHECK_HIP(hipFree(h->blend_w)); h->blend_w = nullptr; h->blend_bytes = 0; }
    hipError_t e = hipMalloc((void**)&h->blend_w, bytes);
    if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B tile blend weights) failed: %s", bytes, hipGetErrorString(e)); return SVI_ERR_OOM; }
    h->blend_bytes = bytes;
    return SVI_OK;
}
}  // namespace

extern "C" svi_status svi_vae_tiled_decode(svi_vae* h, const float* latents, float* video, int32_t T, int32_t hh, int32_t ww,
                                           int32_t size_h, int32_t size_w, int32_t stride_h, int32_t stride_w, svi_stream stream) {
    SVI_REQUIRE(h && latents && video && T > 0 && hh > 0 && ww > 0, "svi_vae_tiled_decode: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE(size_h > 0 && size_w > 0 && stride_h > 0 && stride_w > 0 && stride_h <= size_h && stride_w <= size_w,
                "svi_vae_tiled_decode: tile_size / tile_stride must be positive with stride <= size");
    SVI_TRY(vae_prepare(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int f = 8, To = 4 * T - 3, Ho = hh * f, Wo = ww * f;
    SVI_TRY(ensure_blend(h, (size_t)Ho * Wo * 4));
    const auto hs = tile_starts(hh, size_h, stride_h), wsv = tile_starts(ww, size_w, stride_w);
    // the pool is sized by the largest (first) tile
    const int mh = std::min(size_h, hh), mw = std::min(size_w, ww);
    h->dry = true; h->dry_max = 0;
    SVI_TRY(decode_graph(h, latents, video, T, mh, mw, st));
    h->dry = false;
    SVI_TRY(ensure_pool(h, h->dry_max));
    SVI_CHECK_HIP(hipMemsetAsync(video, 0, (size_t)3 * To * Ho * Wo * 4, st));
    SVI_CHECK_HIP(hipMemsetAsync(h->blend_w, 0, (size_t)Ho * Wo * 4, st));
    svi_status rc = SVI_OK;
    for (auto& a : hs) {
        for (auto& b : wsv) {
            TileWin w{};
            w.Hf = hh; w.Wf = ww; w.h0 = a.first; w.w0 = b.first;
            w.th = std::min(a.second, hh) - a.first; w.tw = std::min(b.second, ww) - b.first;
            w.oHf = Ho; w.oWf = Wo; w.oh0 = a.first * f; w.ow0 = b.first * f;
            w.lb = a.first == 0; w.rb = a.second >= hh; w.tb = b.first == 0; w.bb = b.second >= ww;
            w.border_h = (size_h - stride_h) * f; w.border_w = (size_w - stride_w) * f;
            // the reference writes a `border`-long ramp into the tile's mask: a clipped tile shorter than the ramp is an error there too
            if ((!w.lb || !w.rb) && w.th * f < w.border_h) { svi_set_error("tile of %d rows is shorter than the %d-row blend border", w.th * f, w.border_h); return SVI_ERR_INVALID; }
            if ((!w.tb || !w.bb) && w.tw * f < w.border_w) { svi_set_error("tile of %d columns is shorter than the %d-column blend border", w.tw * f, w.border_w); return SVI_ERR_INVALID; }
            h->slot_used.assign(h->nslots, 0);
            { WinScope ws_(h, &w); rc = decode_graph(h, latents, video, T, w.th, w.tw, st); }
            if (rc != SVI_OK) return rc;
        }
    }
    const long plane = (long)Ho * Wo, planes = 3L * To;
    hipLaunchKernelGGL(tile_finalize_kernel, dim3((unsigned)((planes * plane + 255) / 256)), dim3(256), 0, st, video, h->blend_w, planes, plane, 1);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

extern "C" svi_status svi_vae_tiled_encode(svi_vae* h, const float* video, float* latents, int32_t T, int32_t H, int32_t W,
                                           int32_t size_h, int32_t size_w, int32_t stride_h, int32_t stride_w, svi_stream stream) {
    SVI_REQUIRE(h && video && latents && T > 0 && H > 0 && W > 0, "svi_vae_tiled_encode: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_REQUIRE((T - 1) % 4 == 0 && H % 8 == 0 && W % 8 == 0, "svi_vae_tiled_encode: needs T = 1+4k frames and H, W multiples of 8 (got %d, %d, %d)", T, H, W);
    SVI_REQUIRE(size_h > 0 && size_w > 0 && stride_h > 0 && stride_w > 0 && stride_h <= size_h && stride_w <= size_w &&
                size_h % 8 == 0 && size_w % 8 == 0 && stride_h % 8 == 0 && stride_w % 8 == 0,
                "svi_vae_tiled_encode: tile_size / tile_stride are in pixels, multiples of 8, stride <= size");
    SVI_TRY(vae_prepare(h));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int f = 8, To = (T + 3) / 4, Ho = H / f, Wo = W / f;
    SVI_TRY(ensure_blend(h, (size_t)Ho * Wo * 4));
    const auto hs = tile_starts(H, size_h, stride_h), wsv = tile_starts(W, size_w, stride_w);
    const int mh = std::min(size_h, H), mw = std::min(size_w, W);
    h->dry = true; h->dry_max = 0;
    SVI_TRY(encode_graph(h, video, latents, T, mh, mw, st));
    h->dry = false;
    SVI_TRY(ensure_pool(h, h->dry_max));
    SVI_CHECK_HIP(hipMemsetAsync(latents, 0, (size_t)16 * To * Ho * Wo * 4, st));
    SVI_CHECK_HIP(hipMemsetAsync(h->blend_w, 0, (size_t)Ho * Wo * 4, st));
    svi_status rc = SVI_OK;
    for (auto& a : hs) {
        for (auto& b : wsv) {
            TileWin w{};
            w.Hf = H; w.Wf = W; w.h0 = a.first; w.w0 = b.first;
            w.th = std::min(a.second, H) - a.first; w.tw = std::min(b.second, W) - b.first;
            w.oHf = Ho; w.oWf = Wo; w.oh0 = a.first / f; w.ow0 = b.first / f;
            w.lb = a.first == 0; w.rb = a.second >= H; w.tb = b.first == 0; w.bb = b.second >= W;
            w.border_h = (size_h - stride_h) / f; w.border_w = (size_w - stride_w) / f;
            if ((!w.lb || !w.rb) && w.th / f < w.border_h) { svi_set_error("tile of %d latent rows is shorter than the %d-row blend border", w.th / f, w.border_h); return SVI_ERR_INVALID; }
            if ((!w.tb || !w.bb) && w.tw / f < w.border_w) { svi_set_error("tile of %d latent columns is shorter than the %d-column blend border", w.tw / f, w.border_w); return SVI_ERR_INVALID; }
            h->slot_used.assign(h->nslots, 0);
            { WinScope ws_(h, &w); rc = encode_graph(h, video, latents, T, w.th, w.tw, st); }
            if (rc != SVI_OK) return rc;
        }
    }
    const long plane = (long)Ho * Wo, planes = 16L * To;
    hipLaunchKernelGGL(tile_finalize_kernel, dim3((unsigned)((planes * plane + 255) / 256)), dim3(256), 0, st, latents, h->blend_w, planes, plane, 0);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}


// =================================================================================================================================
// Dance variant: the pose embedder.  pipelines/svi_video_dance.py:255-269 builds
//   Conv3d(3,16,3,p1) SiLU Conv3d(16,16,3,p1) SiLU Conv3d(16,16,3,p1) SiLU Conv3d(16,16,3,s(1,2,2),p1) SiLU
//   Conv3d(16,16,3,s2,p1) SiLU Conv3d(16,16,3,s2,p1) SiLU Conv3d(16,dim,(1,2,2),s(1,2,2))
// (ordinary, non-causal convolutions, fp32) and :527-530 applies it to the pose video with its first frame repeated three more
// times, divided by 255; the result, cast to bf16 and flattened 'b c f h w -> b (f h w) c', is the `add_condition` of every
// conditional forward (:423).  Here: the same seven convolutions on the exact-fp32 MFMA kernel (conv_igemm_kernel, channels-last),
// SiLU fused into the producing convolution, the last one storing bf16 token rows directly.
// =================================================================================================================================
struct svi_pose {
    int device = -1;
    int hidden = 16, dim = 5120;
    ConvW conv[7];
    bool pack_pending = false;
    char* pool = nullptr;
    size_t pool_bytes = 0;
};

namespace {
// pose f32 [3, F, H, W] -> channels-last [F + 3, H, W, 4]: frame 0 three more times in front (dance:529), / 255, 4th channel zero
__global__ void pose_in_kernel(const float* __restrict__ pose, float* __restrict__ out, int F, long hw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)(F + 3) * hw * 4;
    if (i >= n) return;
    const int c = (int)(i & 3);
    const long sp = i >> 2;
    const int t = (int)(sp / hw);
    const long pix = sp - (long)t * hw;
    const int ts = t < 3 ? 0 : t - 3;
    out[i] = c < 3 ? pose[((long)c * F + ts) * hw + pix] / 255.f : 0.f;
}
struct PoseGeom { int T[8], H[8], W[8]; };
PoseGeom pose_geometry(int F, int H, int W) {
    PoseGeom g;
    g.T[0] = F + 3; g.H[0] = H; g.W[0] = W;
    const int st[7] = {1, 1, 1, 1, 2, 2, 1}, ss[7] = {1, 1, 1, 2, 2, 2, 2};
    for (int i = 0; i < 7; ++i) {
        const bool last = i == 6;            // kernel (1,2,2), no padding
        g.T[i + 1] = last ? g.T[i] : (g.T[i] + 2 - 3) / st[i] + 1;
        g.H[i + 1] = last ? (g.H[i] - 2) / 2 + 1 : (g.H[i] + 2 - 3) / ss[i] + 1;
        g.W[i + 1] = last ? (g.W[i] - 2) / 2 + 1 : (g.W[i] + 2 - 3) / ss[i] + 1;
    }
    return g;
}
}  // namespace

extern "C" svi_status svi_pose_create(int32_t hidden, int32_t dim, svi_pose** out) {
    SVI_REQUIRE(out && hidden > 0 && hidden % 4 == 0 && dim > 0, "svi_pose_create: hidden must be a positive multiple of 4, dim positive");
    svi_pose* h = new (std::nothrow) svi_pose();
    if (!h) { svi_set_error("out of host memory"); return SVI_ERR_OOM; }
    h->hidden = hidden; h->dim = dim;
    for (int i = 0; i < 7; ++i) {
        ConvW& c = h->conv[i];
        c.Cin = i == 0 ? 3 : hidden; c.Cout = i == 6 ? dim : hidden;
        c.kt = i == 6 ? 1 : 3; c.kh = c.kw = i == 6 ? 2 : 3;
        c.ldw = (c.Cin + 3) / 4 * 4;
        c.shape = {c.Cout, c.Cin, c.kt, c.kh, c.kw};
    }
    *out = h;
    return SVI_OK;
}

extern "C" svi_status svi_pose_destroy(svi_pose* h) {
    if (!h) return SVI_OK;
    for (auto& c : h->conv) if (c.packed) (void)hipFree(c.packed);
    if (h->pool) (void)hipFree(h->pool);
    delete h;
    return SVI_OK;
}

// name = state-dict key of the reference's nn.Sequential: "<2i>.weight" / "<2i>.bias" for convolution i = 0..6 (dance:270-275 strips
// the "dwpose_embedding." prefix the checkpoint carries).  fp32, as the reference keeps this module.
extern "C" svi_status svi_pose_bind_weight(svi_pose* h, const char* name, const void* dev_ptr, svi_dtype dtype, const int64_t* shape, int32_t rank) {
    SVI_REQUIRE(h && name && dev_ptr && shape, "svi_pose_bind_weight: null argument");
    SVI_REQUIRE(dtype == SVI_F32, "pose embedder parameter '%s' must be fp32", name);
    SVI_REQUIRE(((uintptr_t)dev_ptr % 16) == 0, "pose embedder parameter '%s' is not 16-byte aligned", name);
    char* end = nullptr;
    const long idx = strtol(name, &end, 10);
    if (end == name || *end != '.' || idx < 0 || idx > 12 || (idx & 1) || (strcmp(end, ".weight") != 0 && strcmp(end, ".bias") != 0)) {
        svi_set_error("unknown pose embedder parameter '%s'", name);
        return SVI_ERR_INVALID;
    }
    ConvW& c = h->conv[idx / 2];
    if (strcmp(end, ".bias") == 0) {
        if (!(rank == 1 && shape[0] == c.Cout)) { svi_set_error("shape mismatch for '%s'", name); return SVI_ERR_INVALID; }
        c.b_user = reinterpret_cast<const float*>(dev_ptr);
        return SVI_OK;
    }
    bool ok = rank == 5;
    for (int i = 0; ok && i < 5; ++i) ok = shape[i] == c.shape[i];
    if (!ok) { svi_set_error("shape mismatch for '%s'", name); return SVI_ERR_INVALID; }
    SVI_REQUIRE_DEVICE(h);
    c.w_user = reinterpret_cast<const float*>(dev_ptr);
    const int taps = c.kt * c.kh * c.kw;
    const size_t n = (size_t)taps * c.Cout * c.ldw;
    if (!c.packed) {
        hipError_t e = hipMalloc((void**)&c.packed, n * 4);
        if (e != hipSuccess) { svi_set_error("hipMalloc(packed pose weight) failed: %s", hipGetErrorString(e)); return SVI_ERR_OOM; }
    }
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, c.w_user, c.packed, c.Cout, c.Cin, taps, c.ldw);
    SVI_LAUNCH_CHECK();
    h->pack_pending = true;
    return SVI_OK;
}

extern "C" svi_status svi_pose_check_bound(svi_pose* h) {
    SVI_REQUIRE(h, "null handle");
    for (int i = 0; i < 7; ++i)
        if (!h->conv[i].w_user || !h->conv[i].b_user) { svi_set_error("pose embedder parameter '%d.weight/.bias' was never bound", 2 * i); return SVI_ERR_UNBOUND; }
    return SVI_OK;
}

extern "C" svi_status svi_pose_tokens(svi_pose* h, int32_t F, int32_t H, int32_t W, int32_t* f, int32_t* hh, int32_t* ww) {
    SVI_REQUIRE(h && F > 0 && H >= 16 && W >= 16 && f && hh && ww, "svi_pose_tokens: bad argument");
    const PoseGeom g = pose_geometry(F, H, W);
    *f = g.T[7]; *hh = g.H[7]; *ww = g.W[7];
    return SVI_OK;
}

extern "C" svi_status svi_pose_forward(svi_pose* h, const float* pose, void* out, int32_t F, int32_t H, int32_t W, svi_stream stream) {
    SVI_REQUIRE(h && pose && out && F > 0 && H >= 16 && W >= 16, "svi_pose_forward: bad argument");
    SVI_REQUIRE_DEVICE(h);
    SVI_TRY(svi_pose_check_bound(h));
    if (h->pack_pending) { SVI_CHECK_HIP(hipStreamSynchronize(nullptr)); h->pack_pending = false; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PoseGeom g = pose_geometry(F, H, W);
    // two ping-pong activation buffers sized by the largest layer (the full-resolution ones): [T0, H, W, hidden] fp32
    const size_t slot = (((size_t)g.T[0] * g.H[0] * g.W[0] * (size_t)std::max(h->hidden, 4) * 4) + 255) & ~(size_t)255;
    if (h->pool_bytes < 2 * slot) {
        if (h->pool) { SVI_CHECK_HIP(hipFree(h->pool)); h->pool = nullptr; h->pool_bytes = 0; }
        hipError_t e = hipMalloc((void**)&h->pool, 2 * slot);
        if (e != hipSuccess) { svi_set_error("hipMalloc(%zu B pose embedder activations) failed: %s", 2 * slot, hipGetErrorString(e)); return SVI_ERR_OOM; }
        h->pool_bytes = 2 * slot;
    }
    float* buf[2] = {reinterpret_cast<float*>(h->pool), reinterpret_cast<float*>(h->pool + slot)};
    {
        const long hw = (long)H * W, n = (long)(F + 3) * hw * 4;
        hipLaunchKernelGGL(pose_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pose, buf[0], F, hw);
        SVI_LAUNCH_CHECK();
    }
    const int st_t[7] = {1, 1, 1, 1, 2, 2, 1}, st_s[7] = {1, 1, 1, 2, 2, 2, 2};
    int cur = 0, ld_in = 4;
    for (int i = 0; i < 7; ++i) {
        const ConvW& c = h->conv[i];
        const bool last = i == 6;
        ConvP p{};
        p.in = buf[cur]; p.Ti = g.T[i]; p.Hi = g.H[i]; p.Wi = g.W[i]; p.Cin = (c.Cin + 3) / 4 * 4; p.ld_in = ld_in;
        p.w = c.packed; p.ld_w = c.ldw; p.bias = c.b_user;
        p.kt = c.kt; p.kh = c.kh; p.kw = c.kw; p.st = st_t[i]; p.sh = p.sw = st_s[i];
        p.pt = last ? 0 : 1; p.ph = p.pw = last ? 0 : 1;
        p.To = g.T[i + 1]; p.Ho = g.H[i + 1]; p.Wo = g.W[i + 1]; p.Cout = c.Cout; p.ld_out = c.Cout;
        p.act_silu = last ? 0 : 1;
        if (last) { p.out = nullptr; p.out_bf16 = reinterpret_cast<bf16*>(out); }
        else p.out = buf[cur ^ 1];
        SVI_TRY(launch_conv(p, st));
        cur ^= 1; ld_in = c.Cout;
    }
    return SVI_OK;
}
