// svi_elementwise.hip — HBM-bound row kernels of the DiT block (gfx950).
//
// All of these stream a [rows, dim] bf16 activation once: one 64-lane wave owns one row, each lane
// holds 16-byte (8 x bf16) chunks of it in registers, statistics are reduced across the wave, and
// the row is written once.  Algorithmic traffic = 2 * rows * dim * 2 bytes (read + write); that is
// the figure the roofline fraction of these kernels is quoted against.
//
// Rounding: the reference runs these ops as separate bf16 tensor ops; `rbf()` marks each point
// where it materialises a bf16 tensor so results track it to within accumulation-order noise.
#include "svi_common.h"

#define ROWS_PER_BLOCK 4      // 4 waves / 256 threads per workgroup

// ------------------------------------------------------------------------------------------------
// LayerNorm (no affine | affine) [+ modulate]:  out = ((x-mean)*rstd [*w+b]) [*(1+scale)+shift]
// reference: nn.LayerNorm(eps=1e-6) + modulate(), models/wan_video_dit.py:150-151,331-333,358,370,372
// ------------------------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void ln_mod_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ out,
                                                     int ldo, int rows, int dim, float eps,
                                                     const bf16* __restrict__ w, const bf16* __restrict__ b,
                                                     const float* __restrict__ shift,
                                                     const float* __restrict__ scale1p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = dim >> 3;
    const bf16* xr = x + (size_t)row * ldx;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            bf16x8 t = ld_bf16x8(xr + ci * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = (float)t[j]; s += v[c][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = v[c][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
    bf16* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            const int col = ci * 8;
            bf16x8 o;
            bf16x8 wv, bv;
            if (w) { wv = ld_bf16x8(w + col); bv = ld_bf16x8(b + col); }
            f32x4 sc0, sc1, sh0, sh1;
            if (scale1p) {
                sc0 = *reinterpret_cast<const f32x4*>(scale1p + col); sc1 = *reinterpret_cast<const f32x4*>(scale1p + col + 4);
                sh0 = *reinterpret_cast<const f32x4*>(shift + col); sh1 = *reinterpret_cast<const f32x4*>(shift + col + 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y = (v[c][j] - mean) * rstd;
                if (w) y = y * (float)wv[j] + (float)bv[j];
                y = rbf(y);
                if (scale1p) y = rbf(rbf(y * (j < 4 ? sc0[j & 3] : sc1[j & 3])) + (j < 4 ? sh0[j & 3] : sh1[j & 3]));
                o[j] = (bf16)y;
            }
            st_bf16x8(orow + col, o);
        }
    }
}

// The same arithmetic for rows of whole chunks (dim = 512 MAXC): a wave walks RPW consecutive rows with row i + 1 requested before row i is
// worked on; for MAXC <= 3 (the 1.3B width) the modulation / affine vectors stay in registers instead of being reloaded for every row.
// Bit-identical to ln_mod_kernel.
template <int MAXC, int RPW, bool AFFINE, bool MOD>
__global__ __launch_bounds__(256) void ln_mod_rows_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ out, int ldo, int rows, int dim, float eps,
                                                          const bf16* __restrict__ w, const bf16* __restrict__ b, const float* __restrict__ shift,
                                                          const float* __restrict__ scale1p) {
    constexpr bool HOIST = MAXC <= 3;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int row_end = min(rows, row0 + RPW);
    bf16x8 cur[MAXC], nxt[MAXC];
    bf16x8 wv[HOIST ? MAXC : 1], bv[HOIST ? MAXC : 1];
    f32x4 sc[HOIST ? MAXC : 1][2], sh[HOIST ? MAXC : 1][2];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int col = (lane + 64 * c) * 8;
        cur[c] = ld_bf16x8(x + (size_t)row0 * ldx + col);
        nxt[c] = cur[c];
        if constexpr (HOIST) {
            if constexpr (AFFINE) { wv[c] = ld_bf16x8(w + col); bv[c] = ld_bf16x8(b + col); }
            if constexpr (MOD) {
                sc[c][0] = *reinterpret_cast<const f32x4*>(scale1p + col); sc[c][1] = *reinterpret_cast<const f32x4*>(scale1p + col + 4);
                sh[c][0] = *reinterpret_cast<const f32x4*>(shift + col); sh[c][1] = *reinterpret_cast<const f32x4*>(shift + col + 4);
            }
        }
    }
    for (int row = row0; row < row_end; ++row) {
        if (row + 1 < row_end) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) nxt[c] = ld_bf16x8(x + (size_t)(row + 1) * ldx + (lane + 64 * c) * 8);
        }
        float v[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = (float)cur[c][j]; s += v[c][j]; }
        const float mean = wave_sum(s) / (float)dim;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = v[c][j] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
        bf16* orow = out + (size_t)row * ldo;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = (lane + 64 * c) * 8;
            bf16x8 wc, bc;
            f32x4 sc0, sc1, sh0, sh1;
            if constexpr (HOIST) {
                if constexpr (AFFINE) { wc = wv[c]; bc = bv[c]; }
                if constexpr (MOD) { sc0 = sc[c][0]; sc1 = sc[c][1]; sh0 = sh[c][0]; sh1 = sh[c][1]; }
            } else {
                if constexpr (AFFINE) { wc = ld_bf16x8(w + col); bc = ld_bf16x8(b + col); }
                if constexpr (MOD) {
                    sc0 = *reinterpret_cast<const f32x4*>(scale1p + col); sc1 = *reinterpret_cast<const f32x4*>(scale1p + col + 4);
                    sh0 = *reinterpret_cast<const f32x4*>(shift + col); sh1 = *reinterpret_cast<const f32x4*>(shift + col + 4);
                }
            }
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y = (v[c][j] - mean) * rstd;
                if constexpr (AFFINE) y = y * (float)wc[j] + (float)bc[j];
                y = rbf(y);
                if constexpr (MOD) y = rbf(rbf(y * (j < 4 ? sc0[j & 3] : sc1[j & 3])) + (j < 4 ? sh0[j & 3] : sh1[j & 3]));
                o[j] = (bf16)y;
            }
            st_bf16x8(orow + col, o);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) cur[c] = nxt[c];
    }
}

svi_status svi_launch_ln_mod(const bf16* x, int ldx, bf16* out, int ldo, int rows, int dim, float eps,
                             const bf16* w, const bf16* b, const float* shift, const float* scale1p,
                             hipStream_t st) {
    SVI_REQUIRE(dim % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "layernorm: dim/ld must be multiples of 8 (dim=%d)", dim);
    SVI_REQUIRE(dim <= 8192, "layernorm: dim %d > 8192 unsupported", dim);
    SVI_REQUIRE((w == nullptr) == (b == nullptr), "layernorm: affine weight and bias must come together");
    SVI_REQUIRE((shift == nullptr) == (scale1p == nullptr), "layernorm: shift and scale must come together");
    if (rows <= 0) return SVI_OK;
    dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
    const int nchunk = dim / 8;
    // the DiT's widths (whole chunks per lane) and norm kinds (modulated without affine: norm1 / norm2; affine without modulation: norm3): the
    // multi-row kernel.  The modulation vectors must be 16-byte aligned for its vector loads (the DiT's are).
    if ((nchunk == 64 * 3 || nchunk == 64 * 10) && svi_switches().rms_rows && ((w != nullptr) != (scale1p != nullptr)) &&
        ((((uintptr_t)shift | (uintptr_t)scale1p) & 15) == 0)) {
        constexpr int RPW = 4;
        dim3 grid_r((rows + ROWS_PER_BLOCK * RPW - 1) / (ROWS_PER_BLOCK * RPW));
#define SVI_LN_ROWS(MAXC)                                                                                                                 \
        do {                                                                                                                              \
            if (w) hipLaunchKernelGGL((ln_mod_rows_kernel<MAXC, RPW, true, false>), grid_r, block, 0, st, x, ldx, out, ldo, rows, dim, eps, w, b, shift, scale1p);   \
            else hipLaunchKernelGGL((ln_mod_rows_kernel<MAXC, RPW, false, true>), grid_r, block, 0, st, x, ldx, out, ldo, rows, dim, eps, w, b, shift, scale1p);   \
        } while (0)
        if (nchunk == 64 * 3) SVI_LN_ROWS(3);
        else SVI_LN_ROWS(10);
#undef SVI_LN_ROWS
        SVI_LAUNCH_CHECK();
        return SVI_OK;
    }
    if (nchunk <= 64 * 3)
        hipLaunchKernelGGL(ln_mod_kernel<3>, grid, block, 0, st, x, ldx, out, ldo, rows, dim, eps, w, b, shift, scale1p);
    else if (nchunk <= 64 * 10)
        hipLaunchKernelGGL(ln_mod_kernel<10>, grid, block, 0, st, x, ldx, out, ldo, rows, dim, eps, w, b, shift, scale1p);
    else
        hipLaunchKernelGGL(ln_mod_kernel<16>, grid, block, 0, st, x, ldx, out, ldo, rows, dim, eps, w, b, shift, scale1p);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ------------------------------------------------------------------------------------------------
// RMSNorm over the full model dim (all heads jointly) [+ 3-D RoPE per head], in place.
// reference: RMSNorm.forward models/wan_video_dit.py:192-197; rope_apply :178-183 with the table of
// :161-175 gathered per token at pipelines/svi_video.py:106-110.  The reference rotates in fp64;
// here cos/sin come from a host-built fp64->fp32 table and the 2x2 rotate runs in fp32 — the result
// is rounded to bf16 either way (difference <= 1 bf16 ulp on rounding ties only).
// ------------------------------------------------------------------------------------------------
// blockIdx.y selects one of up to two [rows, dim] operands that sit side by side in a row (q | k of the DiT's QK buffer): operand
// p starts at column p * dim and has its own weight and output scale — one launch normalises both.
template <int MAXC, bool SCATTER>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16* __restrict__ x, int ld, int rows, int dim,
                                                           const bf16* __restrict__ weight, const bf16* __restrict__ weight1, float eps, int use_rope,
                                                           SviRope r, float out_scale, float out_scale1, SviScatter sc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = dim >> 3;
    bf16* xr = x + (size_t)row * ld + (size_t)blockIdx.y * dim;
    if (blockIdx.y) { weight = weight1; out_scale = out_scale1; }
    float v[MAXC][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            bf16x8 t = ld_bf16x8(xr + ci * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = (float)t[j]; ss += v[c][j] * v[c][j]; }
        }
    }
    const float rs = rsqrtf(wave_sum(ss) / (float)dim + eps);
    int pf = 0, ph = 0, pw = 0;
    const float2* tt = nullptr;                  // this token's 64 pairs, when the caller gathered them (SviRope::tab_tok)
    if (use_rope) {
        const int hw = r.h * r.w;
        int tok = row;
        if (r.period > 0) tok %= r.period;
        tok += r.row0;
        if (r.tab_tok) tt = r.tab_tok + (size_t)tok * 64;
        pf = tok / hw;
        const int rem = tok - pf * hw;
        ph = rem / r.w;
        pw = rem - ph * r.w;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            const int col = ci * 8;
            bf16x8 wv = ld_bf16x8(weight + col);
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = rbf(rbf(v[c][j] * rs) * (float)wv[j]);
            bf16x8 o;
            if (use_rope) {
                const int pair0 = (col & 127) >> 1;            // complex-pair index inside the head
                f32x4 t01 = {0.f, 0.f, 0.f, 0.f}, t23 = t01;
                if (tt) {
                    t01 = *reinterpret_cast<const f32x4*>(tt + pair0);
                    t23 = *reinterpret_cast<const f32x4*>(tt + pair0 + 2);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int pi = pair0 + p;
                    float2 cs;
                    if (tt) cs = p == 0 ? make_float2(t01[0], t01[1]) : p == 1 ? make_float2(t01[2], t01[3]) : p == 2 ? make_float2(t23[0], t23[1]) : make_float2(t23[2], t23[3]);
                    else if (pi < r.npf) cs = r.tab_f[pf * r.npf + pi];
                    else if (pi < r.npf + r.nph) cs = r.tab_h[ph * r.nph + (pi - r.npf)];
                    else cs = r.tab_w[pw * r.npw + (pi - r.npf - r.nph)];
                    const float a = y[2 * p], bq = y[2 * p + 1];
                    o[2 * p] = (bf16)((a * cs.x - bq * cs.y) * out_scale);
                    o[2 * p + 1] = (bf16)((a * cs.y + bq * cs.x) * out_scale);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)(y[j] * out_scale);
            }
            if constexpr (SCATTER) {
                const int j = col / sc.Dp, cl = col - j * sc.Dp, gq = cl / sc.Dg, cg = cl - gq * sc.Dg;
                bf16* ob = blockIdx.y ? sc.out1 : sc.out0;
                int rps = rows, rr = row, nbr = 1, smp = 0;
                if (sc.rows_per_sample > 0) { rps = sc.rows_per_sample; nbr = rows / rps; smp = row / rps; rr = row - smp * rps; }
                st_bf16x8(ob + (((size_t)(gq * sc.P + j) * rps + rr) * nbr + smp) * sc.Dg + cg, o);
            } else {
                st_bf16x8(xr + col, o);
            }
        }
    }
}

// The same arithmetic for the shapes the DiT runs (every lane holds MAXC whole chunks: dim = 512 MAXC; RoPE from the per-token table or none;
// in place): a wave walks RPW consecutive rows, requests row i + 1 before it works on row i, and keeps the gain vector in registers — the
// generic kernel above reloads and unpacks it for every row and has one row in flight per wave.  Bit-identical results.
// Q8 (opt-in fp8 QK^T attention): the result rows leave as MX e4m3 bytes + E8M0 block scales (operand blockIdx.y -> q8o.q8 / q8o.k8) instead of bf16 —
// a lane's 8 channels are a quarter of a 32-channel block and 16 lanes one head: the quantiser's own lane layout (mx8_quant8), so the bits are those
// svi_launch_mx8_quantize makes of the bf16 rows this kernel would have written.
template <int MAXC, int RPW, bool ROPE, bool Q8 = false>
__global__ __launch_bounds__(256) void rmsnorm_rope_rows_kernel(bf16* __restrict__ x, int ld, int rows, int dim, const bf16* __restrict__ weight,
                                                                const bf16* __restrict__ weight1, float eps, SviRope r, float out_scale, float out_scale1, SviQk8 q8o = SviQk8{}) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int row_end = min(rows, row0 + RPW);
    bf16* xb = x + (size_t)blockIdx.y * dim;
    if (blockIdx.y) { weight = weight1; out_scale = out_scale1; }
    bf16x8 wv[MAXC], cur[MAXC], nxt[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        wv[c] = ld_bf16x8(weight + (lane + 64 * c) * 8);
        cur[c] = ld_bf16x8(xb + (size_t)row0 * ld + (lane + 64 * c) * 8);
        nxt[c] = cur[c];
    }
    for (int row = row0; row < row_end; ++row) {
        if (row + 1 < row_end) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) nxt[c] = ld_bf16x8(xb + (size_t)(row + 1) * ld + (lane + 64 * c) * 8);
        }
        float v[MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = (float)cur[c][j]; ss += v[c][j] * v[c][j]; }
        const float rs = rsqrtf(wave_sum(ss) / (float)dim + eps);
        const float2* tt = nullptr;
        if (ROPE) {
            int tok = row;
            if (r.period > 0) tok %= r.period;
            tok += r.row0;
            tt = r.tab_tok + (size_t)tok * 64;
        }
        bf16* xr = xb + (size_t)row * ld;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int col = (lane + 64 * c) * 8;
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = rbf(rbf(v[c][j] * rs) * (float)wv[c][j]);
            bf16x8 o;
            if (ROPE) {
                const int pair0 = (col & 127) >> 1;
                const f32x4 t01 = *reinterpret_cast<const f32x4*>(tt + pair0), t23 = *reinterpret_cast<const f32x4*>(tt + pair0 + 2);
                const float cx[4] = {t01[0], t01[2], t23[0], t23[2]}, sy[4] = {t01[1], t01[3], t23[1], t23[3]};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float a = y[2 * p], bq = y[2 * p + 1];
                    o[2 * p] = (bf16)((a * cx[p] - bq * sy[p]) * out_scale);
                    o[2 * p + 1] = (bf16)((a * sy[p] + bq * cx[p]) * out_scale);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)(y[j] * out_scale);
            }
            if constexpr (Q8) {
                float vq[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vq[j] = (float)o[j];
                mx8_quant8(vq, true, row, lane + 64 * c, blockIdx.y ? q8o.k8 : q8o.q8, q8o.ld8, blockIdx.y ? q8o.ks : q8o.qs, blockIdx.y ? q8o.ks_rows : q8o.qs_rows);
            } else {
                st_bf16x8(xr + col, o);
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) cur[c] = nxt[c];
    }
}

// rs[m] = rsqrt(mean_n q[m][n]^2 + eps) from the per-64-column sums of squares the q projection's epilogue left (SviGemmArgs::rowss): groups summed in
// index order; the division and the rsqrt are rmsnorm_rope_kernel's.  flash_cross_kernel applies the normalisation as it reads q.
__global__ __launch_bounds__(256) void row_rs_kernel(const float* __restrict__ rowss, int groups, int ldss, int rows, float dim, float eps, float* __restrict__ rs) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= rows) return;
    float ss = 0.f;
    int g = 0;
    for (; g + 8 <= groups; g += 8) {          // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = rowss[(size_t)(g + i) * ldss + m];
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += v[i];
    }
    for (; g < groups; ++g) ss += rowss[(size_t)g * ldss + m];
    rs[m] = rsqrtf(ss / dim + eps);
}
svi_status svi_launch_row_rs(const float* rowss, int groups, int ldss, int rows, int dim, float eps, float* rs, hipStream_t st) {
    SVI_REQUIRE(rowss && rs && groups > 0 && groups * 64 == dim && ldss >= rows, "row_rs: bad layout (groups %d, dim %d, ldss %d, rows %d)", groups, dim, ldss, rows);
    if (rows <= 0) return SVI_OK;
    hipLaunchKernelGGL(row_rs_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, rowss, groups, ldss, rows, (float)dim, eps, rs);      // (1024 workgroups at the C2 size: every CU busy)
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

bool svi_rmsnorm_rope_q8_ok(int dim, const SviRope* rope) {
    return (dim == 512 * 3 || dim == 512 * 10) && rope && rope->tab_tok && svi_switches().rms_rows;
}
svi_status svi_launch_rmsnorm_rope2_q8(bf16* x, int ld, int rows, int dim, const bf16* weight, const bf16* weight1, float eps, const SviRope* rope, float out_scale,
                                       float out_scale1, hipStream_t st, const SviQk8& out) {
    SVI_REQUIRE(svi_rmsnorm_rope_q8_ok(dim, rope) && weight1 && ld >= 2 * dim && ld % 8 == 0, "rmsnorm -> e4m3: only the DiT's q | k launch (dim %d)", dim);
    SVI_REQUIRE(out.q8 && out.k8 && out.qs && out.ks && out.ld8 >= dim && out.ld8 % 8 == 0 && out.qs_rows >= rows && out.ks_rows >= rows, "rmsnorm -> e4m3: bad output buffers");
    SVI_REQUIRE(rope->npf + rope->nph + rope->npw == 64 && rope->row0 >= 0 && (rope->period > 0 ? (rope->row0 + rope->period <= rope->f * rope->h * rope->w && rows % rope->period == 0) : rope->row0 + rows <= rope->f * rope->h * rope->w),
                "rope grid %dx%dx%d does not cover rows [%d, %d)", rope->f, rope->h, rope->w, rope->row0, rope->row0 + rows);
    if (rows <= 0) return SVI_OK;
    constexpr int RPW = 4;
    dim3 grid_r((rows + ROWS_PER_BLOCK * RPW - 1) / (ROWS_PER_BLOCK * RPW), 2), block(256);
    if (dim == 512 * 3) hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<3, RPW, true, true>), grid_r, block, 0, st, x, ld, rows, dim, weight, weight1, eps, *rope, out_scale, out_scale1, out);
    else hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<10, RPW, true, true>), grid_r, block, 0, st, x, ld, rows, dim, weight, weight1, eps, *rope, out_scale, out_scale1, out);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

svi_status svi_launch_rmsnorm_rope(bf16* x, int ld, int rows, int dim, const bf16* weight, float eps,
                                   const SviRope* rope, float out_scale, hipStream_t st) {
    return svi_launch_rmsnorm_rope2(x, ld, rows, dim, weight, nullptr, eps, rope, out_scale, 1.0f, st);
}

// weight1 != nullptr: a second operand at columns [dim, 2 dim) of every row gets weight1 / out_scale1 in the same launch
svi_status svi_launch_rmsnorm_rope2(bf16* x, int ld, int rows, int dim, const bf16* weight, const bf16* weight1, float eps,
                                    const SviRope* rope, float out_scale, float out_scale1, hipStream_t st, const SviScatter* scatter) {
    SVI_REQUIRE(dim % 8 == 0 && ld % 8 == 0, "rmsnorm: dim/ld must be multiples of 8");
    SVI_REQUIRE(dim <= 8192, "rmsnorm: dim %d > 8192 unsupported", dim);
    if (rows <= 0) return SVI_OK;
    SviRope r{};
    if (rope) {
        r = *rope;
        SVI_REQUIRE(dim % 128 == 0 && r.npf + r.nph + r.npw == 64, "rope needs head_dim 128");
        SVI_REQUIRE(r.row0 >= 0 && (r.period > 0 ? (r.row0 + r.period <= r.f * r.h * r.w && rows % r.period == 0) : r.row0 + rows <= r.f * r.h * r.w),
                    "rope grid %dx%dx%d does not cover rows [%d, %d)", r.f, r.h, r.w, r.row0, r.row0 + rows);
    }
    SVI_REQUIRE(!weight1 || ld >= 2 * dim, "rmsnorm: a second operand needs ld >= 2 dim");
    dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, weight1 ? 2 : 1), block(256);
    const int nchunk = dim / 8;
    const int use = rope ? 1 : 0;
    SviScatter sc{};
    if (scatter) {
        sc = *scatter;
        SVI_REQUIRE(sc.out0 && (!weight1 || sc.out1) && sc.P > 0 && sc.Dp > 0 && sc.Dg > 0 && sc.P * sc.Dp == dim && sc.Dp % sc.Dg == 0 && sc.Dg % 8 == 0,
                    "rmsnorm: bad send layout (P=%d Dp=%d Dg=%d for dim %d)", sc.P, sc.Dp, sc.Dg, dim);
        SVI_REQUIRE(sc.rows_per_sample == 0 || (sc.rows_per_sample > 0 && rows % sc.rows_per_sample == 0),
                    "rmsnorm: %d rows are not whole samples of %d rows", rows, sc.rows_per_sample);
    }
#define SVI_RMS_LAUNCH(MAXC)                                                                                                             \
    do {                                                                                                                                 \
        if (scatter) hipLaunchKernelGGL((rmsnorm_rope_kernel<MAXC, true>), grid, block, 0, st, x, ld, rows, dim, weight, weight1, eps, use, r, out_scale, out_scale1, sc);  \
        else hipLaunchKernelGGL((rmsnorm_rope_kernel<MAXC, false>), grid, block, 0, st, x, ld, rows, dim, weight, weight1, eps, use, r, out_scale, out_scale1, sc);        \
    } while (0)
    // the DiT's shapes (in place, whole chunks per lane, per-token RoPE table or no RoPE): the multi-row kernel
    if (!scatter && (nchunk == 64 * 3 || nchunk == 64 * 10) && (!rope || r.tab_tok) && svi_switches().rms_rows) {
        constexpr int RPW = 4;
        dim3 grid_r((rows + ROWS_PER_BLOCK * RPW - 1) / (ROWS_PER_BLOCK * RPW), weight1 ? 2 : 1);
#define SVI_RMS_ROWS(MAXC)                                                                                                               \
        do {                                                                                                                             \
            if (rope) hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<MAXC, RPW, true>), grid_r, block, 0, st, x, ld, rows, dim, weight, weight1, eps, r, out_scale, out_scale1);   \
            else hipLaunchKernelGGL((rmsnorm_rope_rows_kernel<MAXC, RPW, false>), grid_r, block, 0, st, x, ld, rows, dim, weight, weight1, eps, r, out_scale, out_scale1);       \
        } while (0)
        if (nchunk == 64 * 3) SVI_RMS_ROWS(3);
        else SVI_RMS_ROWS(10);
#undef SVI_RMS_ROWS
        SVI_LAUNCH_CHECK();
        return SVI_OK;
    }
    if (nchunk <= 64 * 3) SVI_RMS_LAUNCH(3);
    else if (nchunk <= 64 * 10) SVI_RMS_LAUNCH(10);
    else SVI_RMS_LAUNCH(16);
#undef SVI_RMS_LAUNCH
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ------------------------------------------------------------------------------------------------
// bf16 transpose through LDS (64x64 tiles): out[c][r] = in[r][c].  Used by the flash_attention seam
// to present V as V^T (the DiT forward never needs it: its V^T comes straight out of a swapped GEMM).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16* __restrict__ in, int ldi, bf16* __restrict__ out,
                                                        int ldo, int rows, int cols) {
    __shared__ bf16 tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ldi + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(size_t)c * ldo + r] = tile[tx][i];
    }
}

svi_status svi_launch_transpose(const bf16* in, int ldi, bf16* out, int ldo, int rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return SVI_OK;
    dim3 grid((cols + 63) / 64, (rows + 63) / 64), block(256);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, in, ldi, out, ldo, rows, cols);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// ------------------------------------------------------------------------------------------------
// CFG combine + flow-match Euler step (pipelines/svi_video.py:410,420; schedulers/flow_match.py:63)
//   v   = uncond + s*(cond-uncond)        (bf16 tensor ops: each result rounded)
//   lat = lat + v*(sigma_next - sigma)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfg_step_kernel(bf16* __restrict__ lat, const bf16* __restrict__ cond,
                                                       const bf16* __restrict__ uncond, int64_t n, float s,
                                                       float dsigma) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float c = (float)cond[i];
        float v = c;
        if (uncond) {
            const float u = (float)uncond[i];
            v = rbf(u + rbf(s * rbf(c - u)));
        }
        lat[i] = (bf16)((float)lat[i] + rbf(v * dsigma));
    }
}

svi_status svi_launch_cfg_step(bf16* lat, const bf16* cond, const bf16* uncond, int64_t n, float s, float dsigma,
                               hipStream_t st) {
    if (n <= 0) return SVI_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cfg_step_kernel, dim3(blocks), dim3(256), 0, st, lat, cond, uncond, n, s, dsigma);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

__global__ __launch_bounds__(256) void add_bf16_kernel(bf16* __restrict__ a, const bf16* __restrict__ b, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        a[i] = (bf16)((float)a[i] + (float)b[i]);
}

// out = bf16(a - b): the TeaCache residual, hidden_states - previous_hidden_states (pipelines/svi_video.py:65)
__global__ __launch_bounds__(256) void sub_bf16_kernel(bf16* __restrict__ out, const bf16* __restrict__ a, const bf16* __restrict__ b, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (bf16)((float)a[i] - (float)b[i]);
}

svi_status svi_launch_sub_bf16(bf16* out, const bf16* a, const bf16* b, int64_t n, hipStream_t st) {
    if (n <= 0) return SVI_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sub_bf16_kernel, dim3(blocks), dim3(256), 0, st, out, a, b, n);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

svi_status svi_launch_add_bf16(bf16* a, const bf16* b, int64_t n, hipStream_t st) {
    if (n <= 0) return SVI_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(blocks), dim3(256), 0, st, a, b, n);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}


// ------------------------------------------------------------------------------------------------
// Frame hand-off of the clip loop, on the device.  The reference turns the decoded video into 8-bit frames
// (SVIVideoPipeline.tensor2video, pipelines/svi_video.py:366-370: ((x + 1) * 127.5).clip(0, 255).astype(uint8), 'C T H W -> T H W C')
// and turns the last frames back into [-1, 1] floats for the next clip's conditioning (BasePipeline.preprocess_image,
// pipelines/base.py:44-45: float32(x) * (2 / 255) - 1, 'H W C -> C H W').  Same fp32 arithmetic here, so the 8-bit frames and the
// conditioning input are bit-identical to the host round trip — without leaving HBM.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void video_to_u8_kernel(const float* __restrict__ v, unsigned char* __restrict__ out, long thw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;           // over T*H*W*3 outputs
    if (idx >= thw * 3) return;
    const long px = idx / 3;
    const int c = (int)(idx - px * 3);
    float y = (v[(long)c * thw + px] + 1.0f) * 127.5f;
    y = fminf(fmaxf(y, 0.0f), 255.0f);
    out[idx] = (unsigned char)(int)y;                                       // astype(uint8): truncation
}
__global__ __launch_bounds__(256) void u8_to_video_kernel(const unsigned char* __restrict__ f, float* __restrict__ out, int n, long hw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;           // over n*3*H*W outputs, [n, 3, H, W]
    if (idx >= (long)n * 3 * hw) return;
    const long px = idx % hw;
    const int c = (int)((idx / hw) % 3);
    const long i = idx / (3 * hw);
    out[idx] = (float)f[(i * hw + px) * 3 + c] * (float)(2.0 / 255.0) - 1.0f;
}

svi_status svi_launch_video_to_u8(const float* video, unsigned char* out, long thw, hipStream_t st) {
    if (thw <= 0) return SVI_OK;
    hipLaunchKernelGGL(video_to_u8_kernel, dim3((unsigned)((thw * 3 + 255) / 256)), dim3(256), 0, st, video, out, thw);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}
svi_status svi_launch_u8_to_video(const unsigned char* frames, float* out, int n, long hw, hipStream_t st) {
    if (n <= 0 || hw <= 0) return SVI_OK;
    hipLaunchKernelGGL(u8_to_video_kernel, dim3((unsigned)(((long)n * 3 * hw + 255) / 256)), dim3(256), 0, st, frames, out, n, hw);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}


// ------------------------------------------------------------------------------------------------
// FP8 weight storage (BASELINE configs[4]).  The reference's "FP8 quantization" (test_svi.py:337,
// vram_management/layers.py:65-71) stores every parameter as float8_e4m3fn and casts it to the computation dtype (bf16) in front of
// every use: W_eff = bf16(e4m3(W)) — arithmetic stays bf16.  e4m3fn -> bf16 is exact (3 mantissa bits, exponents 2^-9 .. 2^8), so
// the cast is done ONCE at bind time: 288 GB of HBM hold the bf16 copy next to the fp8 original, and every forward is then the
// ordinary bf16 forward on exactly the reference's effective weights.
// OCP e4m3fn: 1-4-3, bias 7, no infinities, S.1111.111 = NaN, subnormals m * 2^-9.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fp8_e4m3_to_bf16_kernel(const unsigned char* __restrict__ in, bf16* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned b = in[i];
    const unsigned sign = (b & 0x80u) << 24, e = (b >> 3) & 15u, m = b & 7u;
    unsigned bits;
    if (e == 15u && m == 7u) bits = sign | 0x7fc00000u;                                  // NaN
    else if (e == 0u) bits = sign | __builtin_bit_cast(unsigned, (float)m * 0.001953125f);   // subnormal: m * 2^-9 (exact in fp32)
    else bits = sign | ((e + 120u) << 23) | (m << 20);
    out[i] = (bf16)__builtin_bit_cast(float, bits);                                      // exact: the low 16 bits are zero
}
svi_status svi_launch_fp8_e4m3_to_bf16(const unsigned char* in, bf16* out, int64_t n, hipStream_t st) {
    if (n <= 0) return SVI_OK;
    hipLaunchKernelGGL(fp8_e4m3_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}


// ------------------------------------------------------------------------------------------------
// Sequence-parallel exchanges, receive side (svi_hip/sequence_parallel.py).  The send side needs no kernel: q | k leave the
// RMSNorm+RoPE launch in send order (SviScatter), V^T [D, ldvt] and the attention output [G][L][Dg] already are contiguous per peer.
// ------------------------------------------------------------------------------------------------
// V^T pieces [P(src)][Dp][lds] -> out.  A piece's columns are the source's token rows, branch b (of nb stacked CFG branches) at columns
// [b * Ls, (b + 1) * Ls) (columns past nb * Ls are padding); channel c = g * Dg + cg of the piece goes to row (g * nb + b) * Dg + cg of out
// [G * nb * Dg][L8], column src * Ls + i: per head group the branches' channels are adjacent — 2 x as many heads of one attention launch.
// nb = 1: out [Dp][L8], row c.  VEC elements per thread.
template <int VEC>
__global__ __launch_bounds__(256) void sp_unpack_vt_kernel(const bf16* __restrict__ recv, bf16* __restrict__ out, int P, int Dp, int Ls, int lds, int L8, int nb, int Dg) {
    const int per_row = Ls / VEC;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)P * Dp * nb * per_row;
    if (idx >= n) return;
    const int i = (int)(idx % per_row) * VEC;
    int64_t rc = idx / per_row;
    const int b = (int)(rc % nb); rc /= nb;
    const int c = (int)(rc % Dp), src = (int)(rc / Dp);
    const int g = c / Dg, cg = c - g * Dg;
    const bf16* ip = recv + ((size_t)src * Dp + c) * lds + (size_t)b * Ls + i;
    bf16* op = out + ((size_t)(g * nb + b) * Dg + cg) * L8 + (size_t)src * Ls + i;
    if constexpr (VEC == 8) st_bf16x8(op, ld_bf16x8(ip));
    else if constexpr (VEC == 2) *reinterpret_cast<unsigned*>(op) = *reinterpret_cast<const unsigned*>(ip);
    else *op = *ip;
}
svi_status svi_launch_sp_unpack_vt(const bf16* recv, bf16* out, int P, int Dp, int Ls, int lds, int L8, hipStream_t st, int nb, int Dg) {
    if (Dg <= 0) Dg = Dp;
    SVI_REQUIRE(P > 0 && Dp > 0 && Ls > 0 && nb >= 1 && lds >= nb * Ls && L8 >= P * Ls && Dp % Dg == 0, "sp_unpack_vt: bad sizes");
    const int vec = (Ls % 8 == 0 && lds % 8 == 0 && L8 % 8 == 0) ? 8 : (Ls % 2 == 0 && lds % 2 == 0 && L8 % 2 == 0) ? 2 : 1;
    const int64_t n = (int64_t)P * Dp * nb * (Ls / vec);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (vec == 8) hipLaunchKernelGGL(sp_unpack_vt_kernel<8>, grid, block, 0, st, recv, out, P, Dp, Ls, lds, L8, nb, Dg);
    else if (vec == 2) hipLaunchKernelGGL(sp_unpack_vt_kernel<2>, grid, block, 0, st, recv, out, P, Dp, Ls, lds, L8, nb, Dg);
    else hipLaunchKernelGGL(sp_unpack_vt_kernel<1>, grid, block, 0, st, recv, out, P, Dp, Ls, lds, L8, nb, Dg);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}
// attention output pieces [G][P(src)][Ls][nb][Dg] -> out [nb * Ls][P * G * Dg]: branch b's token rows at [b * Ls, (b + 1) * Ls), source j's head block at
// columns [j*Dp, (j+1)*Dp), Dp = G*Dg
__global__ __launch_bounds__(256) void sp_unpack_out_kernel(const bf16* __restrict__ recv, bf16* __restrict__ out, int P, int G, int Ls, int Dg, int nb) {
    const int cpr = Dg >> 3;                                            // 16-byte chunks per piece row
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)G * P * Ls * nb * cpr;
    if (idx >= n) return;
    const int ch = (int)(idx % cpr);
    int64_t t = idx / cpr;
    const int b = (int)(t % nb); t /= nb;
    const int row = (int)(t % Ls); t /= Ls;
    const int j = (int)(t % P), g = (int)(t / P);
    const int D = P * G * Dg;
    st_bf16x8(out + ((size_t)b * Ls + row) * D + (size_t)j * G * Dg + (size_t)g * Dg + ch * 8, ld_bf16x8(recv + idx * 8));
}
svi_status svi_launch_sp_unpack_out(const bf16* recv, bf16* out, int P, int G, int Ls, int Dg, hipStream_t st, int nb) {
    SVI_REQUIRE(P > 0 && G > 0 && Ls > 0 && Dg > 0 && Dg % 8 == 0 && nb >= 1, "sp_unpack_out: bad sizes");
    const int64_t n = (int64_t)G * P * Ls * nb * (Dg / 8);
    hipLaunchKernelGGL(sp_unpack_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, recv, out, P, G, Ls, Dg, nb);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}


// ------------------------------------------------------------------------------------------------
// Three-way guidance of the talk sampler + Euler step (pipelines/svi_video_talk.py:455-461, schedulers/flow_match.py:63):
//   v = uncond + s_text * (cond - drop_text) + s_audio * (drop_text - uncond)      (bf16 tensor ops, left to right, each rounded)
//   lat = lat + v * (sigma_next - sigma)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfg3_step_kernel(bf16* __restrict__ lat, const bf16* __restrict__ cond, const bf16* __restrict__ uncond,
                                                        const bf16* __restrict__ drop, int64_t n, float st, float sa, float dsigma) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float c = (float)cond[i], u = (float)uncond[i], d = (float)drop[i];
        const float a = rbf(st * rbf(c - d));
        const float b = rbf(sa * rbf(d - u));
        const float v = rbf(rbf(u + a) + b);
        lat[i] = (bf16)((float)lat[i] + rbf(v * dsigma));
    }
}
svi_status svi_launch_cfg3_step(bf16* lat, const bf16* cond, const bf16* uncond, const bf16* drop, int64_t n, float st, float sa, float dsigma,
                                hipStream_t stream) {
    if (n <= 0) return SVI_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cfg3_step_kernel, dim3(blocks), dim3(256), 0, stream, lat, cond, uncond, drop, n, st, sa, dsigma);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}
