// svi_attention.hip — flash-attention forward for the Wan DiT on gfx950 (head_dim 128, unmasked).
//
// Stands in for flash_attention()/F.scaled_dot_product_attention at models/wan_video_dit.py:116-147:
// 70 % of the FLOPs of a DiT block at 81f@832x480 (L = 32760 tokens, 12 heads).
//
// Work split: grid = (ceil(Lq/128), heads).  A 256-thread workgroup owns 128 query rows of one head;
// each of its 4 waves owns 32 query rows and walks the whole key axis in 64-key tiles.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16; lane l -> row/col (l & 31), k-block hi = l >> 5):
//   S^T = K·Q^T   A-operand = K tile rows (keys), B-operand = Q rows.  The accumulator then holds, for ONE
//                 query (l & 31), 16 keys per 32-key block -> the softmax row reduction is 31 in-lane
//                 max/adds plus a single exchange with lane^32.
//   K rows are fed to the MFMA in an order with key bits 2 and 3 swapped.  With that permutation the 8
//   accumulator registers 8s..8s+7 of lane-half hi are the 8 CONSECUTIVE keys 16s + 8hi + 0..7, i.e.
//   exactly the k-block the next MFMA wants from this lane:
//   O^T = V^T·P^T A-operand = V^T rows (channels) x 8 consecutive keys (one ds_read_b128 from the V^T
//                 tile), B-operand = this lane's 8 probabilities packed to bf16.  No cross-lane movement
//                 of P, no transpose reads.  O^T keeps the query on (l & 31), so the online-softmax
//                 rescale of O is lane-local too.
//   V arrives already transposed (V^T [channel][key]): the DiT forward emits it directly from the V
//   projection by swapping the GEMM operands, so no transpose kernel runs on the hot path.
//
// LDS: 2 stages x (K tile [64 keys][128 ch] 16 KiB + V^T tile [128 ch][64 keys] 16 KiB) = 64 KiB ->
// 2 workgroups / CU.  16-byte chunks are XOR-swizzled (K: chunk ^ (row & 15) over a 256-B row; V^T:
// chunk ^ ((row >> 1) & 7) over a 128-B row) so every ds_read_b128 lane group hits 16 distinct slots.
// Global -> LDS through registers: next tile's loads are issued before the MFMAs of the current tile and
// written after them (one barrier per tile).
//
// Algorithmic work: 4 * Lq * Lk * 128 FLOP per head (QK^T + PV, multiply-add = 2).
#include <type_traits>

#include "svi_common.h"

#define QB 128            // query rows per workgroup
#define KB 64             // keys per tile
#define DH 128
#define KT_BYTES (KB * DH * 2)
#define VT_BYTES (DH * KB * 2)

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// MFMA row i of a 32-key block reads key perm23(i): bits 2 and 3 swapped.
__device__ __forceinline__ int perm23(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }

// TAG only names the instantiation (0 = self-attention, 1 = cross-attention) so that profiles tell them apart.
template <int TAG>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(const bf16* __restrict__ Q, int ldq,
                                                           const bf16* __restrict__ K, int ldk,
                                                           const bf16* __restrict__ VT, int ldvt,
                                                           bf16* __restrict__ O, int ldo, int Lq, int Lk,
                                                           float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int head = blockIdx.y;
    const int q_row = blockIdx.x * QB + wave * 32 + l31;
    const bool q_ok = q_row < Lq;

    // ---- Q fragments (B-operand of S^T): 8 k-steps x 8 bf16 ------------------------------------------
    bf16x8 qf[8];
    {
        const bf16* qp = Q + (size_t)(q_ok ? q_row : 0) * ldq + head * DH + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (q_ok) qf[kk] = ld_bf16x8(qp + kk * 16);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[kk][e] = (bf16)0.f;
            }
        }
    }

    // ---- staging assignment -----------------------------------------------------------------------------
    // K tile: 64 rows x 16 chunks; V^T tile: 128 rows x 8 chunks; 1024 chunks each, 4 per thread.
    const int kr = tid >> 4, kc = tid & 15;          // + 16 rows per j
    const int vr = tid >> 3, vc = tid & 7;           // + 32 rows per j
    const bf16* kbase = K + head * DH + kc * 8;
    const bf16* vrow = VT + (size_t)(head * DH) * ldvt;
    u32x4 rk[4], rv[4];
    const int ntiles = (Lk + KB - 1) / KB;

    // Tile loads are unconditional: out-of-range keys are clamped to the last valid key / key-chunk, so the tail
    // tile reads finite duplicates that the -inf mask (scores) and P == 0 (values) remove exactly.
    const int last_key = Lk - 1, last_chunk = (Lk - 1) & ~7;
    auto load_tile = [&](int t) {
        const int key0 = t * KB;
        const int kcol = min(key0 + vc * 8, last_chunk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = min(key0 + kr + 16 * j, last_key);
            rk[j] = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * ldk);
            rv[j] = *reinterpret_cast<const u32x4*>(vrow + (size_t)(vr + 32 * j) * ldvt + kcol);
        }
    };
    auto store_tile = [&](int buf) {
        char* Ks = smem + buf * (KT_BYTES + VT_BYTES);
        char* Vs = Ks + KT_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<u32x4*>(Ks + k_off(kr + 16 * j, kc)) = rk[j];
            *reinterpret_cast<u32x4*>(Vs + v_off(vr + 32 * j, vc)) = rv[j];
        }
    };

    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int krow = perm23(l31);

    // One 64-key tile: S^T = K Q^T, online softmax, O^T += V^T P^T.  MASKED is a compile-time flag so that the
    // key-range test exists only in the peeled last tile (as a runtime `if` the compiler if-converts it into ~110
    // predicated VALU ops per tile).
    auto tile = [&](int t, int cur, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char* Ks = smem + cur * (KT_BYTES + VT_BYTES);
        const char* Vs = Ks + KT_BYTES;
        f32x16 s[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[tt][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + k_off(32 * tt + krow, 2 * kk + hi));
                s[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[tt], 0, 0, 0);
            }
        }
        // register r of s[tt] is key  t*64 + 32*tt + 16*(r>>3) + 8*hi + (r&7)
        if (MASKED) {
            const int kb = t * KB + 8 * hi;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + 32 * tt + 16 * (r >> 3) + (r & 7) >= Lk) s[tt][r] = -INFINITY;
        }
        // ---- online softmax (per query = per (l & 31); the other 32 keys live in lane ^ 32) -----------
        float m8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) m8[r] = fmaxf(fmaxf(s[0][r], s[0][r + 8]), fmaxf(s[1][r], s[1][r + 8]));
        float mx = fmaxf(fmaxf(fmaxf(m8[0], m8[1]), fmaxf(m8[2], m8[3])), fmaxf(fmaxf(m8[4], m8[5]), fmaxf(m8[6], m8[7])));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // Rescale O and l only when some row's running max actually grew in this tile (wave-uniform branch).
        // This is exact, not a threshold: when no max grows alpha == exp2(0) == 1 for every row.
        if (__any(mx > m_run)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        const float mneg = -m_run * scale_log2e;
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
        bf16x8 pf[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[tt][r], scale_log2e, mneg));
                ps[r & 3] += p;
                pf[tt][r >> 3][r & 7] = (bf16)p;
            }
        l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
        // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + v_off(32 * d + l31, 4 * tt + 2 * sb + hi));
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[tt][sb], o[d], 0, 0, 0);
                }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t + 1 < ntiles; ++t) {
        const int cur = t & 1;
        load_tile(t + 1);
        tile(t, cur, std::false_type{});
        store_tile(cur ^ 1);
        __syncthreads();
    }
    if (Lk & (KB - 1)) tile(ntiles - 1, (ntiles - 1) & 1, std::true_type{});
    else tile(ntiles - 1, (ntiles - 1) & 1, std::false_type{});

    // ---- normalise and store: lane holds O[q_row][32 d + (r&3) + 8 (r>>2) + 4 hi] --------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        bf16* op = O + (size_t)q_row * ldo + head * DH + 4 * hi;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (bf16)(o[d][rg * 4 + e] * inv);
                *reinterpret_cast<bf16x4*>(op + 32 * d + 8 * rg) = pk;
            }
    }
}

svi_status svi_launch_flash(const bf16* Q, int ldq, const bf16* K, int ldk, const bf16* VT, int ldvt, bf16* O,
                            int ldo, int Lq, int Lk, int num_heads, hipStream_t st) {
    SVI_REQUIRE(Lq > 0 && Lk > 0 && num_heads > 0, "attention: bad sizes Lq=%d Lk=%d heads=%d", Lq, Lk, num_heads);
    SVI_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "attention: leading dims must be multiples of 8");
    SVI_REQUIRE(ldvt >= ((Lk + 7) / 8) * 8, "attention: V^T leading dim %d < keys rounded up to 8", ldvt);
    SVI_REQUIRE(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)VT % 16) == 0 &&
                    ((uintptr_t)O % 8) == 0, "attention: operands must be 16-byte aligned");
    static bool attr_set = false;
    const int lds = 2 * (KT_BYTES + VT_BYTES);
    if (!attr_set) {
        SVI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_fwd_kernel<0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SVI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_fwd_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    dim3 grid((Lq + QB - 1) / QB, num_heads), block(256);
    if (Lq == Lk)
        hipLaunchKernelGGL(flash_fwd_kernel<0>, grid, block, lds, st, Q, ldq, K, ldk, VT, ldvt, O, ldo, Lq, Lk, scale_log2e);
    else
        hipLaunchKernelGGL(flash_fwd_kernel<1>, grid, block, lds, st, Q, ldq, K, ldk, VT, ldvt, O, ldo, Lq, Lk, scale_log2e);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}
