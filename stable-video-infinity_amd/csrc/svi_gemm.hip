// svi_gemm.hip — bf16 MFMA GEMM for nn.Linear on gfx950:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// Both operands are K-contiguous ("NT"), which is the natural layout of an activation matrix and of a
// PyTorch Linear weight, and exactly what a v_mfma_f32_32x32x16_bf16 fragment wants: lane l holds 8
// consecutive k of row (l & 31), k-block (l >> 5).
//
// Tile: 128(M) x 128(N) x 64(K) per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA
// tiles, 64 accumulator VGPRs).  The MFMA is issued with the WEIGHT fragment as A-operand and the
// ACTIVATION fragment as B-operand, so a lane's accumulator holds 4 consecutive n of ONE activation
// row m = (l & 31): the epilogue then packs 8-byte row pieces instead of 2-byte column pieces.
//
// LDS: two stages x (A tile 16 KiB + W tile 16 KiB) = 64 KiB -> 2 workgroups per CU.  Rows are
// 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 so that the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct 16-byte slots of the 256-byte bank row
// (MI355X_MICROARCH.md §LDS; cdna_hip_programming.md T2).  Global->LDS goes through registers
// (issue the next tile's loads before the MFMAs, write them after; one barrier per K tile).
//
// Epilogue: y = bf16(acc + bias) in registers -> staged through LDS as a bf16 [128][136] tile ->
// re-read row-contiguously (16 B per lane, 256 B per row segment) -> activation / gate / residual ->
// coalesced 16-byte stores.  Residual may alias C (x += gate * y in place).
//
// Workgroup ids are remapped so that the 8 XCDs each own a contiguous band of tiles (per-XCD L2 keeps
// the shared A row-panel hot; cdna_hip_programming.md T1, bijective form).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <atomic>
#include <algorithm>

#include "svi_common.h"
#include <cstdio>

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (BM * BK * 2)      // 16 KiB per operand tile
#define CS_LD 136                      // bf16 elements per staged C row (272 B: 16-B aligned, odd*16)

__device__ __forceinline__ int lds_tile_off(int row, int chunk) {   // byte offset inside a [128][64] bf16 tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// SviGemmArgs::rowss — sum of squares of row m's rounded results over the aligned 64-column group that holds columns [n, n + 8): this lane's 8 columns in
// order, then the xor-1/2/4 tree over the group's 8 lanes (lane & 7 == the chunk's position in the group in every tiled kernel's read-back loop).
__device__ __forceinline__ void gemm_rowss_store(const SviGemmArgs& g, const bf16x8& o, int m, int n, int lane_chunk) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)o[e]; ss += f * f; }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 4);
    if ((lane_chunk & 7) == 0) g.rowss[(size_t)(n >> 6) * g.ldss + m] = ss;
}

// PF = K tiles of global loads in flight (a register ring of PF slots, the K loop unrolled by PF so that every slot is a fixed register set and the compiler
// counts vmcnt itself).  PF = 1 is the loop as rounds 1-6 had it (two workgroups per CU share a SIMD's issue slots).  PF = 4 serves launches of at most one
// workgroup per CU — the C1-size step, where M = 2560 rows make 240 tiles of a projection: a lone wave per SIMD covers only 512 cycles of matrix work per
// K tile, a quarter of an L2 round trip, and the PF = 1 loop ran at that latency (1.2 us per K tile; ffn2 at C1: 169 us for 70 GFLOP).  Same MFMA and k
// order in both: bit-identical.
#ifndef SVI_GEMM_DEEP_PF
#define SVI_GEMM_DEEP_PF 4
#endif
// NT = 256: four waves as 2 x 2, 64 x 64 per wave.  NT = 512 (with PF = 4): eight waves as 2 x 4, 64 x 32 per wave — the same tile, the same LDS image, the same
// k order per element (bit-identical), two waves per SIMD from ONE workgroup: while one waits for its fragments or its staging writes the other multiplies.
template <int PF, int NT = 256>
__global__ __launch_bounds__(NT, (PF == 1 && NT == 256) ? 2 : 1) void gemm_bf16_nt_kernel(SviGemmArgs g, int tiles_m, int tiles_n) {
    constexpr int WNC = NT / 128;           // wave columns (2 or 4)
    constexpr int NBW = 8 / WNC;            // 16-column blocks per wave (4 or 2)
    constexpr int LJ = 1024 / NT;           // 16-byte chunks of an operand tile per thread (4 or 2)
    constexpr int LR = NT / 8;              // tile rows one pass of the staging assignment covers (32 or 64)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WNC, wn = wave % WNC;
    const int l15 = lane & 15, g4 = lane >> 4;

    // XCD-aware, bijective remap of the linear workgroup id
    const int nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
    const int tile_m = swz / tiles_n, tile_n = swz - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (g.W2 && n0 >= g.n_split) {                          // q | k side by side: this tile's columns belong to the second matrix
        g.W = g.W2 - (size_t)g.n_split * g.ldw;
        g.bias = g.bias2 ? g.bias2 - g.n_split : nullptr;
    }

    // global -> register staging assignment: 1024 16-byte chunks per operand tile, LJ per thread
    const int ld_row = tid >> 3;            // + LR*j
    const int ld_chunk = tid & 7;
    const bf16* a_ptr[LJ];
    const bf16* w_ptr[LJ];
    bool a_ok[LJ], w_ok[LJ];
#pragma unroll
    for (int j = 0; j < LJ; ++j) {
        const int r = ld_row + LR * j;
        a_ok[j] = (m0 + r) < g.M;
        w_ok[j] = (n0 + r) < g.N;
        a_ptr[j] = g.A + (size_t)(a_ok[j] ? (m0 + r) : 0) * g.lda + ld_chunk * 8;
        w_ptr[j] = g.W + (size_t)(w_ok[j] ? (n0 + r) : 0) * g.ldw + ld_chunk * 8;
    }
    const int nk = (g.K + BK - 1) / BK;
    u32x4 ra[PF][LJ], rw[PF][LJ];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tile = [&](u32x4 (&xa)[LJ], u32x4 (&xw)[LJ], int kt) {
        if constexpr (PF > 1) {          // K % 64 == 0 (the launcher's condition): no k tail, and a row past the edge reads row 0 (its results are never stored) —
#pragma unroll                           // no predicate, no branch around a load: the compiler's vmcnt arithmetic keeps PF - 1 tiles in flight
            for (int j = 0; j < LJ; ++j) {
                xa[j] = *reinterpret_cast<const u32x4*>(a_ptr[j] + kt * BK);
                xw[j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + kt * BK);
            }
            return;
        }
        const int kbase = kt * BK + ld_chunk * 8;
        const bool kin = kbase < g.K;          // K % 8 == 0: a chunk is entirely inside or outside
#pragma unroll
        for (int j = 0; j < LJ; ++j) {
            xa[j] = (a_ok[j] && kin) ? *reinterpret_cast<const u32x4*>(a_ptr[j] + kt * BK) : zero4;
            xw[j] = (w_ok[j] && kin) ? *reinterpret_cast<const u32x4*>(w_ptr[j] + kt * BK) : zero4;
        }
    };
    auto store_tile = [&](const u32x4 (&xa)[LJ], const u32x4 (&xw)[LJ], int buf) {
        char* As = smem + buf * 2 * STAGE_BYTES;
        char* Ws = As + STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < LJ; ++j) {
            const int off = lds_tile_off(ld_row + LR * j, ld_chunk);
            *reinterpret_cast<u32x4*>(As + off) = xa[j];
            *reinterpret_cast<u32x4*>(Ws + off) = xw[j];
        }
    };

    f32x4 acc[NBW][4];                      // [nb][mb]: 16 x 16 blocks, lane l holds n = 4 (l >> 4) + e of row m = l & 15
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < PF; ++s) load_tile(ra[s], rw[s], s);          // (PF > 1: the launcher guarantees nk >= PF; PF = 1: nk >= 1)
    store_tile(ra[0], rw[0], 0);
    __syncthreads();

    // one K tile: slot s of the ring held tile kt (in LDS since the step before; tile 0: the prologue) and takes tile kt + PF; tile kt + 1 moves to LDS behind the MFMAs
    auto kstep = [&](u32x4 (&la)[LJ], u32x4 (&lw)[LJ], const u32x4 (&sa)[LJ], const u32x4 (&sw_)[LJ], int kt, bool ld, bool st) {
        const int cur = kt & 1;
        if (ld) load_tile(la, lw, kt + PF);
        const char* As = smem + cur * 2 * STAGE_BYTES;
        const char* Ws = As + STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {          // two k-steps of 32: lane group g4 holds k = 8 g4 .. 8 g4 + 7 of the step
            bf16x8 xa[4], wb[NBW];
#pragma unroll
            for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const bf16x8*>(As + lds_tile_off(wm * 64 + i * 16 + l15, 4 * kk + g4));
#pragma unroll
            for (int i = 0; i < NBW; ++i) wb[i] = *reinterpret_cast<const bf16x8*>(Ws + lds_tile_off(wn * (16 * NBW) + i * 16 + l15, 4 * kk + g4));
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nb], xa[mb], acc[nb][mb], 0, 0, 0);
        }
        if (st) store_tile(sa, sw_, cur ^ 1);
        __syncthreads();
    };
    // steady state: every step loads and stores — no branch between a load and the wait that counts it, so PF - 1 tiles stay in flight (a conditional
    // load makes the compiler's vmcnt arithmetic assume the shorter path: vmcnt(0) every step, which is what the PF = 1 loop pays by construction)
    int kt0 = 0;
    for (; kt0 + 2 * PF <= nk; kt0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) kstep(ra[s], rw[s], ra[(s + 1) % PF], rw[(s + 1) % PF], kt0 + s, true, true);
    }
    for (; kt0 < nk; kt0 += PF) {                 // the last PF .. 2 PF - 1 tiles
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int kt = kt0 + s;
            if (kt < nk) kstep(ra[s], rw[s], ra[(s + 1) % PF], rw[(s + 1) % PF], kt, kt + PF < nk, kt + 1 < nk);
        }
    }

    // ---- epilogue part 1: y = bf16(acc + bias) -> LDS [128 m][CS_LD] bf16 --------------------------
    // acc[nb][mb][e] is C[m = m0 + wm*64 + mb*16 + l15][n = n0 + wn*(16 NBW) + nb*16 + 4*g4 + e]
    bf16* Cs = reinterpret_cast<bf16*>(smem);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int nl = wn * (16 * NBW) + nb * 16 + 4 * g4;
        float bn[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bias && !g.bias_along_m) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bn[e] = (n0 + nl + e < g.N) ? (float)g.bias[n0 + nl + e] : 0.f;
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ml = wm * 64 + mb * 16 + l15;
            float bm = 0.f;
            if (g.bias && g.bias_along_m) bm = (m0 + ml < g.M) ? (float)g.bias[m0 + ml] : 0.f;
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16)(acc[nb][mb][e] + ((g.bias && g.bias_along_m) ? bm : bn[e]));
            *reinterpret_cast<bf16x4*>(Cs + ml * CS_LD + nl) = pk;
        }
    }
    __syncthreads();

    // ---- epilogue part 2: row-contiguous read-back, activation / gate / residual, coalesced store ----
#pragma unroll
    for (int it = 0; it < 2048 / NT; ++it) {
        const int id = tid + NT * it;
        const int ml = id >> 4, cc = id & 15;
        const int m = m0 + ml, n = n0 + cc * 8;
        if (m >= g.M || n >= g.N) continue;
        bf16x8 yv = *reinterpret_cast<const bf16x8*>(Cs + ml * CS_LD + cc * 8);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (float)yv[e];
        const bool full = (n + 8 <= g.N);
        if (g.epi == SVI_EPI_BIAS_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = gelu_tanh_f(y[e]);
        } else if (g.epi == SVI_EPI_BIAS_GELU_ERF) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = gelu_erf_f(y[e]);
        } else if (g.epi == SVI_EPI_BIAS_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
        } else if (g.epi == SVI_EPI_BIAS_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], 0.f);
        } else if (g.epi == SVI_EPI_BIAS_GATE_RES) {
            const bf16* rp = g.res + (size_t)m * g.ldres + n;
            float rv[8];
            if (full) {
                bf16x8 t = ld_bf16x8(rp);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = (float)t[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = (n + e < g.N) ? (float)rp[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = y[e];
                if (g.gate) t = rbf(((n + e < g.N) ? g.gate[n + e] : 0.f) * t);
                y[e] = rv[e] + t;
            }
        }
        bf16* cp = g.C + (size_t)m * g.ldc + n;
        if (full) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)y[e];
            st_bf16x8(cp, o);
            if (g.rowss) gemm_rowss_store(g, o, m, n, cc);          // (N % 64 == 0: a group's 8 lanes are all here)
        } else {
            for (int e = 0; e < 8 && n + e < g.N; ++e) cp[e] = (bf16)y[e];
        }
    }
}


// =================================================================================================
// Large-problem kernel: 256(M) x 256(N) x 64(K) per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per
// wave = 4x2 MFMA tiles, 128 accumulator VGPRs), ONE workgroup per CU.
//
// Why a second kernel: with the 128^2 tile a K step gives a wave only 16 MFMAs (512 cycles) of cover for the
// next tile's global loads, so the loop runs at L2/HBM latency, not at MFMA rate (measured 530-610 TFLOP/s).
// Here a K step is 32 MFMAs per wave and two waves share a SIMD -> 2048 cycles of matrix work per K tile,
// and the operand tiles are fetched by LDS-DMA (no staging VGPRs, no ds_write pass; cdna_hip_programming.md §5 "glds vs register staging").
// The main loop itself is gemm_bf16_nt_256e_kernel below; its predecessors (rounds 1-3: the lockstep "v2" / "v3" loops with one barrier per
// K tile, a persistent variant, a 256 x 192 tile on the v3 loop) were bit-identical and slower and are gone from the source — their numbers:
// profiles/r2e_gemm_persistent_ab.txt, profiles/r4b_gemm_ab.txt.
//
// LDS image: per stage A tile [256 rows][128 B] + W tile [256][128 B] = 64 KiB, two stages = 128 KiB.
// LDS-DMA writes lane-linear (wave-uniform base + lane*16), so the bank swizzle sits on the SOURCE address:
// LDS slot (row r, position p) receives global 16-byte chunk p ^ ((r >> 1) & 7) of row r; the fragment read
// applies the same XOR (rule 21: source permutation == read permutation, destination linear).  A wave
// instruction covers 8 whole 128-byte rows, so the permutation stays inside full cache lines.
//
// Epilogue as in the 128^2 kernel, the staged C tile is [256][264] bf16 (132 KiB, the LDS is otherwise idle).
// Tried and dropped (bit-identical, measured on MI355X, tools/gemm_ab.py):  BK = 32 with four LDS stages and counted
// vmcnt (3 tiles in flight): -5 % (twice the barriers, DMA latency was not the limiter);  a persistent kernel that defers
// the epilogue of tile i into the K loop of tile i+1 (64 packed-bf16 registers, permlane-paired 16-byte stores): -5..-13 %
// (register spills, and the row-scattered stores compete with the LDS-DMA for the address path).  The epilogue itself was the
// real loss (see gemm256_epilogue_t): after its rewrite it costs 2-9 us per tile, mostly the 128 KiB of stores.
// Also tried: two independent 4-wave workgroups per CU on 128 x 256 x 32 tiles (three-stage half-tile ring, the v3 barrier
// placement) so that one workgroup's epilogue falls under the other's main loop: bit-identical, but 840-940 TFLOP/s on every
// shape against 970-1260 here — half-tiles double the barriers and DMA instructions per MFMA and raise the L2 -> LDS traffic by
// half, which costs more than the hidden epilogue returns.
// Tile order: 8 XCD bands (bijective), inside a band groups of GM row panels (4 for narrow N, 5 for N = 8960: measured, tools/gemm_gm.py) walk the
// column panels, so the 32 tiles an XCD runs at once are a ~5 x 6 block: ~12 operand panels for 32 tiles in its L2.  That geometry bounds the L2 hit
// rate at 1 - 12/64 = 0.81 (measured 0.75-0.76, profiles/r4z_gemm_ffn*_pmc.json): 0.85 would need reuse ACROSS rounds, and a round's row panels
// (5 x 786 KB at K = 1536) already fill the 4 MB L2.
// =================================================================================================
#define TM 256
#define TN 256
#define TN3 192
#ifndef SVI_GEMM_DEFAULT_256
#define SVI_GEMM_DEFAULT_256 260      // which schedule of the 256^2 tile runs by default: 260 = two phases per K tile (clusters of 32 MFMAs; the faster one since the 16x16x32 instruction: q|k +6 %, ffn +1 %, profiles/r6v_gemm_phases_ab.txt), 259 = four; SVI_GEMM_KERNEL overrides per process
#endif
#define T_STAGE (TM * BK * 2)          // 32 KiB per operand tile
#define C2_LD 264
#define LDS256_BYTES (TM * C2_LD * 2)  // 135168 >= 4 * T_STAGE

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Shared epilogue of the 256^2 kernels: y = bf16(acc + bias) staged through LDS as a [256][C2_LD] bf16 tile, read back row-
// contiguously (512 B per row), activation / gate / residual applied, 16-byte coalesced stores.  All LDS reads of the main
// loop must be complete (barrier) before this is called.
//
// Where its time went (epilogue ablations, tools/gemm_stagger.py; ffn1 = 17.5 tiles per CU): of 1042 us, 384 us were epilogue
// — and none of that was the global stores.  175 us were 128 dependent 2-byte bias loads per thread (one memory round trip each),
// 209 us the read-back loop: one `switch (epi)` chain, four bounds tests and an `s_waitcnt lgkmcnt(0)` per 16-byte chunk, sixteen
// times per thread.  Now: the epilogue kind is a template parameter (one uniform branch per tile), bias values come in as eight
// 8-byte vectors up front, and an interior tile (all but the last row panel) takes a path without bounds tests in which the
// LDS reads of four chunks are in flight together.
// abl (timing ablations): 1 = no epilogue at all, 2 = no global stores, 3 = stop after the LDS staging writes (results wrong);
// 4 = non-temporal stores (results unchanged).
template <int EPI>
__device__ __forceinline__ void gemm256_apply(float (&y)[8], const u32x4& res, const float (&gatev)[8], bool has_gate) {
    if constexpr (EPI == SVI_EPI_BIAS_GELU_TANH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = gelu_tanh_f(y[e]);
    } else if constexpr (EPI == SVI_EPI_BIAS_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = gelu_erf_f(y[e]);
    } else if constexpr (EPI == SVI_EPI_BIAS_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
    } else if constexpr (EPI == SVI_EPI_BIAS_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = fmaxf(y[e], 0.f);
    } else if constexpr (EPI == SVI_EPI_BIAS_GATE_RES) {
        const bf16x8 t = __builtin_bit_cast(bf16x8, res);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = y[e];
            if (has_gate) v = rbf(gatev[e] * v);
            y[e] = (float)t[e] + v;
        }
    }
}

// (mx8_quant8 — eight consecutive elements of a row -> e4m3 bytes + the block's E8M0 code — lives in svi_common.h: the RMSNorm + RoPE kernel shares it)

// MI x NI: 32x32 accumulator blocks per wave along M / N (wave grid (256 / 32 MI) x (TNV / 32 NI)); TNV: tile width.  The 256^2 kernels are
// <4, 2, 256>; the 256 x 192 kernel <2, 3, 192> (read-back threads whose column chunk lies past the tile's 192 columns sit idle).
// NT: threads of the workgroup (512: eight waves, 16 rows per read-back iteration; 256: the four-wave kernel, 8 rows per iteration)
// MF: the MFMA the accumulators come from.  32: f32x16 acc[NI][MI], lane l holds n = 8 rg + 4 (l >> 5) + e (r = 4 rg + e) of row m = l & 31 (the MX fp8 kernel
// and the four-wave experiments);  16: f32x4 acc[2 NI][2 MI] of v_mfma_f32_16x16x32_bf16, lane l holds n = 4 (l >> 4) + e of row m = l & 15 (the bf16 kernels).
template <int EPI, int MI = 4, int NI = 2, int TNV = TN, bool Q8OUT = false, int NT = 512, int MF = 32, class ACC>
__device__ __forceinline__ void gemm256_epilogue_t(const SviGemmArgs& g, ACC& acc, char* smem, int m0, int n0, int tid,
                                                   int wm, int wn, int l31, int hi, int abl) {
    constexpr int RPI = NT / 32, ITS = TM / RPI;          // rows per read-back iteration, iterations
    const bool interior = (m0 + TM <= g.M) && (n0 + TNV <= g.N);
    // gate·y + residual: the 16 residual chunks this thread will need are requested NOW (64 registers, the accumulators
    // are about to die) so that their latency runs under the LDS round trip; the column block (and so the gate values)
    // is the same for all 16 chunks of a thread.
    u32x4 resv[ITS];
    float gatev[8];
    const int ecc = tid & 31, er = tid >> 5, en = n0 + ecc * 8;
    const bool active = TNV == TN || ecc * 8 < TNV;          // this thread's column chunk belongs to the tile
    const bool efull = en + 8 <= g.N;
    if constexpr (EPI == SVI_EPI_BIAS_GATE_RES) {
        if (!active) {
        } else if (interior) {
            const bf16* rp = g.res + (size_t)(m0 + er) * g.ldres + en;
#pragma unroll
            for (int it = 0; it < ITS; ++it) resv[it] = *reinterpret_cast<const u32x4*>(rp + (size_t)(RPI * it) * g.ldres);
        } else {
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int m = m0 + er + RPI * it;
                if (m < g.M && efull) resv[it] = *reinterpret_cast<const u32x4*>(g.res + (size_t)m * g.ldres + en);
            }
        }
        if (!active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gatev[e] = 0.f;
        } else if (g.gate && efull && (((uintptr_t)g.gate & 15) == 0)) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(g.gate + en), g1 = *reinterpret_cast<const f32x4*>(g.gate + en + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gatev[e] = g0[e]; gatev[4 + e] = g1[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) gatev[e] = (g.gate && en + e < g.N) ? g.gate[en + e] : 0.f;
        }
    }
    // ---- part 1: y = bf16(acc + bias) -> LDS [256 m][C2_LD] bf16 ---------------------------------------------------------
    const bool bias_n = g.bias && !g.bias_along_m, bias_m = g.bias && g.bias_along_m;
    const bool bias_vec = bias_n && (n0 + TNV <= g.N) && (((uintptr_t)g.bias & 7) == 0);
    bf16* Cs = reinterpret_cast<bf16*>(smem);
    if constexpr (MF == 16) {
        const int l15 = tid & 15, g4 = (tid & 63) >> 4;
        u32x2 bnq[2 * NI];
        float bmq[2 * MI];
#pragma unroll
        for (int nb = 0; nb < 2 * NI; ++nb) {
            const int n = n0 + wn * (32 * NI) + nb * 16 + 4 * g4;
            if (bias_vec) {
                bnq[nb] = *reinterpret_cast<const u32x2*>(g.bias + n);
            } else {
                unsigned short h4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) h4[e] = (bias_n && n + e < g.N) ? reinterpret_cast<const unsigned short*>(g.bias)[n + e] : (unsigned short)0;
                bnq[nb][0] = (unsigned)h4[0] | ((unsigned)h4[1] << 16);
                bnq[nb][1] = (unsigned)h4[2] | ((unsigned)h4[3] << 16);
            }
        }
#pragma unroll
        for (int mb = 0; mb < 2 * MI; ++mb) {
            const int m = m0 + wm * (32 * MI) + mb * 16 + l15;
            bmq[mb] = (bias_m && m < g.M) ? (float)g.bias[m] : 0.f;
        }
#pragma unroll
        for (int nb = 0; nb < 2 * NI; ++nb) {
            const int nl = wn * (32 * NI) + nb * 16 + 4 * g4;
            const unsigned b01 = bnq[nb][0], b23 = bnq[nb][1];
            const float bn4[4] = {__builtin_bit_cast(float, b01 << 16), __builtin_bit_cast(float, b01 & 0xffff0000u),
                                  __builtin_bit_cast(float, b23 << 16), __builtin_bit_cast(float, b23 & 0xffff0000u)};
#pragma unroll
            for (int mb = 0; mb < 2 * MI; ++mb) {
                const int ml = wm * (32 * MI) + mb * 16 + l15;
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (bf16)(acc[nb][mb][e] + (bn4[e] + bmq[mb]));
                *reinterpret_cast<bf16x4*>(Cs + ml * C2_LD + nl) = pk;
            }
        }
    } else {
    u32x2 bnp[NI][4];
    float bmv[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int n = n0 + wn * (32 * NI) + ni * 32 + 8 * rg + 4 * hi;
            if (bias_vec) {
                bnp[ni][rg] = *reinterpret_cast<const u32x2*>(g.bias + n);
            } else {
                unsigned short h4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) h4[e] = (bias_n && n + e < g.N) ? reinterpret_cast<const unsigned short*>(g.bias)[n + e] : (unsigned short)0;
                bnp[ni][rg][0] = (unsigned)h4[0] | ((unsigned)h4[1] << 16);
                bnp[ni][rg][1] = (unsigned)h4[2] | ((unsigned)h4[3] << 16);
            }
        }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * (32 * MI) + mi * 32 + l31;
        bmv[mi] = (bias_m && m < g.M) ? (float)g.bias[m] : 0.f;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ml = wm * (32 * MI) + mi * 32 + l31;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = wn * (32 * NI) + ni * 32 + 8 * rg + 4 * hi;
                const unsigned b01 = bnp[ni][rg][0], b23 = bnp[ni][rg][1];
                const float bv[4] = {__builtin_bit_cast(float, b01 << 16) + bmv[mi], __builtin_bit_cast(float, b01 & 0xffff0000u) + bmv[mi],
                                     __builtin_bit_cast(float, b23 << 16) + bmv[mi], __builtin_bit_cast(float, b23 & 0xffff0000u) + bmv[mi]};
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (bf16)(acc[ni][mi][rg * 4 + e] + bv[e]);
                *reinterpret_cast<bf16x4*>(Cs + ml * C2_LD + nl) = pk;
            }
        }
    }
    }
    __syncthreads();
    if (abl == 3) return;
    if (!active) return;                 // (no barrier follows)

    // ---- part 2: row-contiguous read-back (512 B per row), activation / gate / residual, coalesced store ---------------------
    const bool has_gate = g.gate != nullptr;
    if (interior) {
        const bf16* lp = Cs + er * C2_LD + ecc * 8;
        bf16* cp = g.C + (size_t)(m0 + er) * g.ldc + en;
#pragma unroll
        for (int b = 0; b < ITS / 4; ++b) {
            u32x4 yv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) yv[j] = *reinterpret_cast<const u32x4*>(lp + (RPI * (4 * b + j)) * C2_LD);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int it = 4 * b + j;
                const bf16x8 t = __builtin_bit_cast(bf16x8, yv[j]);
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (float)t[e];
                gemm256_apply<EPI>(y, resv[it], gatev, has_gate);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)y[e];
                if constexpr (Q8OUT) {                 // quantise the rounded result instead of storing it (every lane of the wave is here: interior tile)
                    float vq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) vq[e] = (float)o[e];
                    mx8_quant8(vq, true, m0 + er + RPI * it, en >> 3, g.q8, g.ldq8, g.q8s, g.q8_sc_rows);
                    continue;
                }
                if (abl == 2) { if (y[0] == 123.456f && y[7] == 1.f) cp[0] = o[3]; }
                // abl 4: non-temporal stores.  Alone the GEMM gains (attn_o 153 -> 140 us, ffn1 861 -> 841 us: the tile no longer
                // pushes operand panels out of L2), but in the block the next kernel (LN / the next GEMM) then finds its input in
                // HBM instead of L2 / MALL and gives the time back: step 470.3 vs 469.7 ms.  Ordinary stores stay.
                else if (abl == 4) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(cp + (size_t)(RPI * it) * g.ldc));
                else st_bf16x8(cp + (size_t)(RPI * it) * g.ldc, o);
                if constexpr (EPI == SVI_EPI_BIAS && !Q8OUT) {
                    if (g.rowss) gemm_rowss_store(g, o, m0 + er + RPI * it, en, ecc);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int ml = er + RPI * it;
        const int m = m0 + ml, n = en;
        if (m >= g.M || n >= g.N) continue;
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(Cs + ml * C2_LD + ecc * 8);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (float)yv[e];
        u32x4 rvv = resv[it];
        if constexpr (EPI == SVI_EPI_BIAS_GATE_RES) {
            if (!efull) {
                bf16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (n + e < g.N) ? g.res[(size_t)m * g.ldres + n + e] : (bf16)0.f;
                rvv = __builtin_bit_cast(u32x4, t);
            }
        }
        gemm256_apply<EPI>(y, rvv, gatev, has_gate);
        if constexpr (Q8OUT) {                         // (N % 256 == 0: column chunks are whole; rows past M skipped above in whole 32-lane halves)
            float vq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vq[e] = (float)(bf16)y[e];
            mx8_quant8(vq, true, m, n >> 3, g.q8, g.ldq8, g.q8s, g.q8_sc_rows);
            continue;
        }
        bf16* cp = g.C + (size_t)m * g.ldc + n;
        if (efull) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)y[e];
            st_bf16x8(cp, o);
            if constexpr (EPI == SVI_EPI_BIAS && !Q8OUT) {
                if (g.rowss) gemm_rowss_store(g, o, m, n, ecc);
            }
        } else {
            for (int e = 0; e < 8 && n + e < g.N; ++e) cp[e] = (bf16)y[e];
        }
    }
}

template <int MI = 4, int NI = 2, int TNV = TN, bool Q8OUT = false, int NT = 512, int MF = 32, class ACC>
__device__ __forceinline__ void gemm256_epilogue(const SviGemmArgs& g, ACC& acc, char* smem, int m0, int n0, int tid,
                                                 int wm, int wn, int l31, int hi, int abl = 0) {
    if (abl == 1) {
        if (acc[0][0][0] == 123.456f) g.C[0] = (bf16)acc[1][MI - 1][3];      // keep the accumulators alive
        return;
    }
    switch (g.epi) {        // uniform: one scalar branch per tile
        case SVI_EPI_BIAS_GELU_TANH: gemm256_epilogue_t<SVI_EPI_BIAS_GELU_TANH, MI, NI, TNV, Q8OUT, NT, MF>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl); break;
        case SVI_EPI_BIAS_GATE_RES:  gemm256_epilogue_t<SVI_EPI_BIAS_GATE_RES, MI, NI, TNV, Q8OUT, NT, MF>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl); break;
        case SVI_EPI_BIAS_GELU_ERF:  gemm256_epilogue_t<SVI_EPI_BIAS_GELU_ERF, MI, NI, TNV, Q8OUT, NT, MF>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl); break;
        case SVI_EPI_BIAS_SILU:      gemm256_epilogue_t<SVI_EPI_BIAS_SILU, MI, NI, TNV, Q8OUT, NT, MF>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl); break;
        case SVI_EPI_BIAS_RELU:      gemm256_epilogue_t<SVI_EPI_BIAS_RELU, MI, NI, TNV, Q8OUT, NT, MF>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl); break;
        default:                     gemm256_epilogue_t<SVI_EPI_BIAS, MI, NI, TNV, Q8OUT, NT, MF>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl); break;
    }
}

typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_t;
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"     // LDS addresses are 32-bit; the host pass sees 64-bit pointers

// =================================================================================================
// 256-row tiles, STAGGERED main loop ("256e"): same LDS image, same operand swizzle, same MFMA (32x32x16, weight fragment as A operand), same
// k order per element and the same epilogue as the round 1-3 kernels it replaces (and as the 128^2 kernel above) — so the same bits — on another schedule.
//
// Why: the round-3 ("v3") loop kept the SIMD's two waves in lockstep (one barrier per K tile; both waves read fragments, issued LDS-DMA and multiplied at
// the same points of their streams) and held 0.57-0.64 of the matrix pipe in-clock (PMC, profiles/r3n_gemm_ffn1_pmc.txt).  Here, as in the
// schedule cdna_hip_programming.md measures at ~0.75 in-clock on this part ("256^2 8-phase template"), the two wave rows of the workgroup
// — the waves that share a SIMD — run ONE BARRIER APART: on every SIMD one wave is inside an MFMA cluster (raised priority) while its
// partner reads the fragments of its next cluster from LDS and issues its share of the LDS-DMA — matrix beside memory on every interval —
// and the DMA stream is never drained: counted vmcnt, raw s_barrier, half-tiles always in flight.
// Measured (tools/gemm_ab.py, bit-identical to the v3 kernel, profiles/r4b_gemm_ab.txt): ffn2 1199 -> 1364 TFLOP/s, ffn1 1058 -> 1145,
// attn-out 866 -> 936, 8192^3 1168 -> 1329.
//
// A phase = { L: fragment reads + LDS-DMA issue ; wait for the own reads (+ counted vmcnt) ; barrier ; M: one MFMA cluster ; barrier }.
// Tile widths: TNW = 256 (waves 2 x 4, 128 x 64 per wave: MI = 4, NI = 2) or 192 (waves 4 x 2, 64 x 96 per wave: MI = 2, NI = 3 — the tile
// that fills the rounds of a sequence-parallel shard's N = 1536 GEMMs).  Two schedules of a K tile (64 deep):
//   PH == 2 (both widths) — clusters of 4 NI MH MFMAs (MH = MI / 2 row blocks):
//     phase 0   L: W(all NI), A(upper MH row blocks) of tile t    DMA: W of tile t+1    M: acc[.][lower MH] += W x A(lower)   (A(lower) was read in phase 1 of tile t-1)
//     phase 1   L: A(lower MH row blocks) of tile t+1             DMA: A of tile t+2    M: acc[.][upper MH] += W x A(upper)
//   PH == 4 (TNW = 256) — clusters of 8 MFMAs:
//     phase 0   L: W(n0), W(n1) of tile t          DMA: W rows   0..127 of tile t+1     M: acc[n0][m0,m1] += W(n0) x A(m0,m1)
//     phase 1   L: A(m2,m3) of tile t              DMA: W rows 128..255 of tile t+1     M: acc[n1][m0,m1] += W(n1) x A(m0,m1)
//     phase 2   L: A(m0,m1) of tile t+1            DMA: A rows   0..127 of tile t+2     M: acc[n1][m2,m3] += W(n1) x A(m2,m3)
//     phase 3   L: -                               DMA: A rows 128..255 of tile t+2     M: acc[n0][m2,m3] += W(n0) x A(m2,m3)
// Each cluster walks the tile's four k-steps in order, so an accumulator sees k in increasing order, as in every other kernel of this file.  The minimum of fragment reads per
// K tile (4 (MI + NI)), and every LDS read of tile t is over one phase into it, which is what lets tile t+2 stream into tile t's buffer
// with only two LDS buffers.  PH == 2 halves the barriers per MFMA and measures within 1 % of PH == 4 (-3 % on K = 1536, +1 % on K = 8960).
// Safety of the hand-off (cdna_hip_programming.md "read a staged buffer one phase after the wait that retires it"):
//   * a wave waits for its OWN fragment reads (lgkmcnt(0)) BEFORE the barrier that ends its L section: what a later DMA overwrites has
//     been read by everyone once that barrier is passed;
//   * a wave waits for its own DMA shares with a counted vmcnt before that same barrier (vector memory operations complete in issue order on
//     this family: vmcnt(n) = everything but the n newest has landed) and the data is first read one phase later, i.e. behind a barrier
//     BOTH wave rows have passed after their waits.
// Row OOB: the operands are read through buffer descriptors sized to the matrix, rows past M / N read zeros (their results are never stored).
// Tried on this loop and dropped (round 4, bit-identical, profiles/r4d_gemm_persistent_ab.txt): a PERSISTENT form — one workgroup per CU walking
// its tiles, the last K tile staging K tiles 0 and 1 of the next output tile, a wave-private epilogue in the 32 KiB above the operand buffers, the
// one-barrier stagger kept across the boundary.  q/k/v 154 vs 148 us, attn-out 172 vs 154, ffn1 861 vs 786, ffn2 696 vs 654: the private epilogue
// (four 32-row pieces through 4 KiB, each a wave-local LDS round trip) and the boundary K tiles' scalar bookkeeping cost more than the ~3.7 us of
// launch + prologue per tile they remove — the same verdict round 2 reached on the lockstep loop.
// =================================================================================================
#define E_HALF 16384           // an A half-tile: 128 rows x 128 B
#define E_BUF 65536            // one K tile: A [256][128 B] | W [<= 256][128 B]
#define E_BAR() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define E_WAIT(s) asm volatile(s ::: "memory")

template <int PH, int TNW>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_256e_kernel(SviGemmArgs g, int tiles_m, int tiles_n, int GM, int abl_arg) {
    static_assert((TNW == 256 && (PH == 2 || PH == 4)) || (TNW == 192 && PH == 2), "unsupported schedule");
    constexpr int MI = TNW == 256 ? 4 : 2, NI = TNW == 256 ? 2 : 3, MH = MI / 2;
    constexpr int WP = TNW / 64;                          // LDS-DMA pieces of a W tile per wave (A: 4, two per half-tile)
#ifdef SVI_ABLATIONS
    const int abl = abl_arg;          // timing ablations, variant builds only (tools/gemm_epi_abl.py): 1-4 see gemm256_epilogue_t; 5 = one K tile only; 6 = one K tile, no epilogue
#else
    constexpr int abl = 0;
    (void)abl_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lds0 = (int)(size_t)(lptr_t)smem;          // 0: the dynamic region is all the LDS this kernel has (the buffer flip below is an XOR)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // the wave row that shares SIMDs with the other one: waves w and w + 4
    const int wm = TNW == 256 ? wave >> 2 : wave >> 1, wn = TNW == 256 ? wave & 3 : wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
    const int group = swz / (GM * tiles_n);
    const int first_m = group * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int in_group = swz - group * GM * tiles_n;
    const int tile_n = in_group / gm;
    const int tile_m = first_m + (in_group - tile_n * gm);
    const int m0 = tile_m * TM, n0 = tile_n * TNW;
    if (g.W2 && n0 >= g.n_split) {                          // q | k side by side: this tile's columns belong to the second matrix
        g.W = g.W2 - (size_t)g.n_split * g.ldw;
        g.bias = g.bias2 ? g.bias2 - g.n_split : nullptr;
    }
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, (int)(((unsigned)(g.M - 1) * (unsigned)g.lda + (unsigned)g.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.W), 0, (int)(((unsigned)(g.N - 1) * (unsigned)g.ldw + (unsigned)g.K) * 2u), 0x00020000);
    // a wave's DMA instruction covers 8 rows x 128 B; piece p = j * 8 + wave of an operand tile holds its rows j * 64 + wave * 8 + lane / 8.
    // The bank swizzle sits on the SOURCE chunk (LDS-DMA writes lane-linear; header of this section).
    const int r8 = wave * 8 + (lane >> 3);
    const int c8 = (lane & 7) ^ ((r8 >> 1) & 7);
    const int a_vo = (r8 * g.lda + c8 * 8) * 2, w_vo = (r8 * g.ldw + c8 * 8) * 2;
    const unsigned a_so0 = (unsigned)m0 * (unsigned)g.lda * 2u, w_so0 = (unsigned)n0 * (unsigned)g.ldw * 2u;
    const unsigned a_j = 64u * (unsigned)g.lda * 2u, w_j = 64u * (unsigned)g.ldw * 2u;
    const int nk = abl >= 5 ? 1 : g.K / BK;
    // pieces j0 .. j0 + NJ - 1 of the A (op 0) / W (op 1) tile of K tile kt -> buffer b
    auto stage = [&](auto opc, auto j0c, auto njc, int kt, int b) {
        constexpr int op = decltype(opc)::value, j0 = decltype(j0c)::value, NJ = decltype(njc)::value;
        const unsigned so = (op ? w_so0 : a_so0) + (unsigned)kt * (BK * 2);
        char* dst = smem + b * E_BUF + op * (2 * E_HALF) + wave * 1024;
#pragma unroll
        for (int j = j0; j < j0 + NJ; ++j) {
            if constexpr (op == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(dst + j * 8192), 16, a_vo, (int)(so + (unsigned)j * a_j), 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + j * 8192), 16, w_vo, (int)(so + (unsigned)j * w_j), 0, 0);
        }
    };
    std::integral_constant<int, 0> OPA, J0;
    std::integral_constant<int, 1> OPW;
    std::integral_constant<int, 2> J2;
    std::integral_constant<int, 4> J4;
    std::integral_constant<int, WP> JW;

    // The multiply is v_mfma_f32_16x16x32_bf16 (round 6; tools/probe/run_mfma_power_probe.py: on random operands the part's power limit holds a stream of
    // v_mfma_f32_32x32x16_bf16 at 1.73 GHz = 1750 TFLOP/s and a stream of 16x16x32 at 1.99 GHz = 1996, same nominal rate): the 32-row / 32-column blocks the
    // schedule below is written in are two 16-row blocks each, a K tile is two k-steps of 32.  Lane l of a fragment read holds row (l & 15), k = 8 (l >> 4) .. + 7
    // of the step = 16-byte chunk 4 kk + (l >> 4) of the 128-byte row; accumulator block [nb][mb] holds n = 4 (l >> 4) + e of row m = l & 15.
    f32x4 acc[2 * NI][2 * MI];
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
        for (int j = 0; j < 2 * MI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int l15 = lane & 15, g4 = lane >> 4;
    int a_addr[2], w_addr[2];          // this lane's fragment of k-step kk: 16-row block 0 of the wave's rows (A) / columns (W), buffer 0
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_addr[kk] = lds0 + lds_tile_off(wm * (32 * MI) + l15, 4 * kk + g4);
        w_addr[kk] = lds0 + 2 * E_HALF + lds_tile_off(wn * (32 * NI) + l15, 4 * kk + g4);
    }
    u32x4 xlo[2 * MH][2], xup[2 * MH][2], xw[2 * NI][2];          // A fragments of the lower / upper MH 32-row blocks, W fragments of the NI 32-column blocks
    auto read_a = [&](u32x4 (&x)[2 * MH][2], int blk0) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2 * MH; ++i) x[i][kk] = *(lds_u32x4_t)(a_addr[kk] + (2 * blk0 + i) * 16 * 128);
    };
    auto read_w = [&](auto n0c, auto nnc) {
        constexpr int nb0 = decltype(n0c)::value, NN = decltype(nnc)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 2 * nb0; i < 2 * (nb0 + NN); ++i) xw[i][kk] = *(lds_u32x4_t)(w_addr[kk] + i * 16 * 128);
    };
    // acc[n][2 mb0 + i] += W(n) x A(i) for the 16-column blocks n of 32-column blocks [nb0, nb0 + NN), the 16-row blocks i of MH 32-row blocks, over the tile's two k-steps
    auto cluster = [&](auto n0c, auto nnc, int mb0, const u32x4 (&xa)[2 * MH][2]) {
        constexpr int nb0 = decltype(n0c)::value, NN = decltype(nnc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int n = 2 * nb0; n < 2 * (nb0 + NN); ++n)
#pragma unroll
                for (int i = 0; i < 2 * MH; ++i)
                    acc[n][2 * mb0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xw[n][kk]), __builtin_bit_cast(bf16x8, xa[i][kk]), acc[n][2 * mb0 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    std::integral_constant<int, 0> N0;
    std::integral_constant<int, 1> N1;
    std::integral_constant<int, NI> NALL;
    auto flip = [&](int (&ad)[2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) ad[kk] ^= E_BUF;
    };
    // one K tile.  SW: tile t+1 exists (its W is staged here, its lower A fragments are read here);  SA: tile t+2 exists (its A is staged here)
    auto ktile = [&](int t, auto swc, auto sac) {
        constexpr bool SW = decltype(swc)::value, SA = decltype(sac)::value;
        const int b = t & 1;
        if constexpr (PH == 2) {
            // ---- phase 0
            read_w(N0, NALL);
            read_a(xup, MH);
            if constexpr (SW) {
                stage(OPW, J0, JW, t + 1, b ^ 1);
                if constexpr (WP == 4) E_WAIT("s_waitcnt vmcnt(4) lgkmcnt(0)");      // A of tile t+1 has landed (this wave's shares); its W may still fly
                else E_WAIT("s_waitcnt vmcnt(3) lgkmcnt(0)");
            } else {
                E_WAIT("s_waitcnt lgkmcnt(0)");
            }
            E_BAR();
            cluster(N0, NALL, 0, xlo);
            E_BAR();
            // ---- phase 1
            flip(a_addr); flip(w_addr);
            if constexpr (SW) read_a(xlo, 0);
            if constexpr (SA) {
                stage(OPA, J0, J4, t + 2, b);
                E_WAIT("s_waitcnt vmcnt(4) lgkmcnt(0)");                              // W of tile t+1 has landed; A of tile t+2 in flight
            } else if constexpr (SW) {
                E_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");
            }
            E_BAR();
            cluster(N0, NALL, MH, xup);
            E_BAR();
        } else {
            // ---- phase 0
            read_w(N0, NALL);
            if constexpr (SW) stage(OPW, J0, J2, t + 1, b ^ 1);
            E_WAIT("s_waitcnt lgkmcnt(0)");
            E_BAR();
            cluster(N0, N1, 0, xlo);
            E_BAR();
            // ---- phase 1
            read_a(xup, MH);
            if constexpr (SW) {
                stage(OPW, J2, J2, t + 1, b ^ 1);
                E_WAIT("s_waitcnt vmcnt(4) lgkmcnt(0)");                              // A of tile t+1 has landed; its W half-tiles may still fly
            } else {
                E_WAIT("s_waitcnt lgkmcnt(0)");
            }
            E_BAR();
            cluster(N1, N1, 0, xlo);
            E_BAR();
            // ---- phase 2
            flip(a_addr);
            if constexpr (SW) read_a(xlo, 0);
            if constexpr (SA) stage(OPA, J0, J2, t + 2, b);
            E_WAIT("s_waitcnt lgkmcnt(0)");
            E_BAR();
            cluster(N1, N1, MH, xup);
            E_BAR();
            // ---- phase 3
            if constexpr (SA) {
                stage(OPA, J2, J2, t + 2, b);
                E_WAIT("s_waitcnt vmcnt(4)");                                         // W of tile t+1 has landed; A of tile t+2 in flight
            } else if constexpr (SW) {
                E_WAIT("s_waitcnt vmcnt(0)");
            }
            E_BAR();
            cluster(N0, N1, MH, xup);
            E_BAR();
            flip(w_addr);
        }
    };
    std::true_type YES;
    std::false_type NO;

    // prologue: tile 0 whole, the A tile of tile 1 behind it
    stage(OPA, J0, J4, 0, 0);
    stage(OPW, J0, JW, 0, 0);
    if (nk > 1) {
        stage(OPA, J0, J4, 1, 1);
        E_WAIT("s_waitcnt vmcnt(4)");
    } else {
        E_WAIT("s_waitcnt vmcnt(0)");
    }
    E_BAR();
    read_a(xlo, 0);
    if (grp == 1) E_BAR();                                 // the second wave row runs one barrier behind the first from here on
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(t, YES, YES);
    if (nk >= 2) { ktile(t, YES, NO); ++t; }
    ktile(t, NO, NO);
    if (grp == 0) E_BAR();
#pragma unroll
    for (int n = 0; n < 2 * NI; ++n)
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) asm volatile("" : "+v"(acc[n][i]));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // MFMA result -> VALU read
    __syncthreads();                                       // every wave is done with the operand buffers: the C tile may overwrite them
    gemm256_epilogue<MI, NI, TNW, false, 512, 16>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, abl == 5 ? 0 : abl == 6 ? 1 : abl);
}

// =================================================================================================
// Round 6 experiment, NOT part of the product build (-DSVI_GEMM_EXPERIMENTS, tools/build_gemm_variant.py; kinds 264 / 265 of SVI_GEMM_KERNEL): the vendor library's
// shape for these problems — a 256 x 256 x 64 tile on FOUR waves, 128 x 128 per wave, one wave per SIMD with the whole 512-register file (hipBLASLt's
// MT256x256x64 kernel: 256 threads; 1230-1290 TFLOP/s at K = 1536 and 1485 at K = 8960 against 1156-1195 / 1380 for the eight-wave kernel above, same box,
// profiles/r6a_yardstick.txt).  Both forms below are bit-identical to the kernels above (tools/gemm_ab.py, tools/gemm_race_screen.py clean) and SLOWER:
//                                   q|k|v    attn-out   ffn1     ffn2 (K = 8960)      [TFLOP/s, profiles/r6e_gemm_ab_w4.txt, r6f_gemm_ab_w4r.txt]
//   eight waves (kind 259)          1148     1029       1163     1368
//   w4,  LDS-DMA fed (264)          1070      898       1069     1199
//   w4r, fed through registers (265) 825      763        927     1076
//   w4 with its global loads REMOVED 1096     1025       1237     1504      (results wrong: the multiply-and-LDS-read ceiling of the structure = the vendor's rate)
// Reading: the four-wave structure itself reaches the vendor's rate, but a lone wave pays for every memory instruction it issues with matrix-pipe idle time
// — a `buffer_load ... lds` holds the wave's issue ~60 cycles against the 32 an MFMA covers, sixteen per K tile and wave; moving them earlier, spreading them
// (1 per 2 MFMAs) or replacing them by buffer_load + ds_write_b128 through two register sets (a ~3000-cycle prefetch window, 32 memory instructions per
// tile) changes nothing or makes it worse.  In the eight-wave kernel the SIMD's other wave multiplies during those issue slots, which is why it stays the
// product kernel.  What the vendor's hand-written loop does differently was not found from outside.
// =================================================================================================
#ifdef SVI_GEMM_EXPERIMENTS
// =================================================================================================
// 256 x 256 x 64 tile on FOUR waves ("w4", round 6): one wave per SIMD, each owning a 128 x 128 quarter of the tile (MI = NI = 4: sixteen 32x32 accumulator
// blocks = 256 registers, the kernel runs at the full 512-register budget of a lone wave) — the shape the vendor library picks for these problems
// (tools/yardstick.py + rocprofv3: hipBLASLt's MT256x256x64 kernel is 256 threads per workgroup; 1230-1290 TFLOP/s at K = 1536, 1485 at K = 8960 against
// 1156-1195 / 1380 for the eight-wave kernel above, same box, same job).  Why it can be faster where the chip is POWER-limited (tools/attn_stats.py: the same
// instruction stream runs 0.83 of peak on zeros and 0.56 on Gaussian operands): a 128 x 128 wave tile feeds 64 MFMAs from 32 fragment reads per K tile
// (0.5 per MFMA; the 128 x 64 wave tiles above need 0.75), there is ONE barrier per K tile instead of eight, and no priority flips.
// Same LDS image (two 64 KiB stages, A | W tiles of 128-byte rows, source-side XOR swizzle), the same MFMA and operand roles and the same k order per
// accumulator as every other kernel of this file: the same bits.
// One K tile t (buffer b = t & 1), per wave — fragments are double-buffered in registers, k-step kk + 1 is read from LDS while k-step kk multiplies:
//   k-step 0   read k-step 1 of tile t          DMA: the wave's 8 W pieces of tile t+1 -> buffer b^1      16 MFMAs
//   k-step 1   read k-step 2                                                                           16 MFMAs
//   k-step 2   read k-step 3                                                                           16 MFMAs
//   s_waitcnt lgkmcnt(0) (every read of buffer b by this wave is complete) vmcnt(0) (its pieces of tile t+1 have landed), s_barrier
//   k-step 3   read k-step 0 of tile t+1 (b^1)  DMA: the wave's 8 A pieces of tile t+2 -> buffer b        16 MFMAs
// A DMA piece is requested at least two k-steps (>= 1000 cycles) before the wait that retires it; a buffer is overwritten only behind the barrier that
// follows its last read; a staged tile is read only behind the barrier that follows every wave's wait for its pieces (cdna_hip_programming.md "read a
// staged buffer one phase after the wait that retires it").  Within a k-step the source alternates one MFMA with one LDS read / one DMA piece, pinned
// by sched_barrier, so the matrix pipe is never left waiting behind a burst of memory instructions (a lone wave has no partner to cover one).
// =================================================================================================
#define W4_KSTEP_FENCE() __builtin_amdgcn_sched_barrier(0)
__global__ __launch_bounds__(256, 1) void gemm_bf16_nt_w4_kernel(SviGemmArgs g, int tiles_m, int tiles_n, int GM) {
    constexpr int MI = 4, NI = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lds0 = (int)(size_t)(lptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
    const int group = swz / (GM * tiles_n);
    const int first_m = group * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int in_group = swz - group * GM * tiles_n;
    const int tile_n = in_group / gm;
    const int tile_m = first_m + (in_group - tile_n * gm);
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    if (g.W2 && n0 >= g.n_split) {                          // q | k side by side: this tile's columns belong to the second matrix
        g.W = g.W2 - (size_t)g.n_split * g.ldw;
        g.bias = g.bias2 ? g.bias2 - g.n_split : nullptr;
    }
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, (int)(((unsigned)(g.M - 1) * (unsigned)g.lda + (unsigned)g.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.W), 0, (int)(((unsigned)(g.N - 1) * (unsigned)g.ldw + (unsigned)g.K) * 2u), 0x00020000);
    // a wave's DMA instruction covers 8 rows x 128 B; piece j (0..7) of this wave holds rows j * 32 + wave * 8 + lane / 8 of an operand tile; the bank
    // swizzle sits on the SOURCE chunk (as in the kernels above; (row >> 1) & 7 does not depend on j)
    const int r8 = wave * 8 + (lane >> 3);
    const int c8 = (lane & 7) ^ ((r8 >> 1) & 7);
    const int a_vo = (r8 * g.lda + c8 * 8) * 2, w_vo = (r8 * g.ldw + c8 * 8) * 2;
    const unsigned a_so0 = (unsigned)m0 * (unsigned)g.lda * 2u, w_so0 = (unsigned)n0 * (unsigned)g.ldw * 2u;
    const unsigned a_j = 32u * (unsigned)g.lda * 2u, w_j = 32u * (unsigned)g.ldw * 2u;
    const int nk = g.K / BK;
    auto dma = [&](auto opc, int j, int kt, int b) {      // piece j of the A (op 0) / W (op 1) tile of K tile kt -> buffer b
        constexpr int op = decltype(opc)::value;
        char* dst = smem + b * E_BUF + op * (2 * E_HALF) + j * 4096 + wave * 1024;
        if constexpr (op == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)dst, 16, a_vo, (int)(a_so0 + (unsigned)kt * (BK * 2) + (unsigned)j * a_j), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)dst, 16, w_vo, (int)(w_so0 + (unsigned)kt * (BK * 2) + (unsigned)j * w_j), 0, 0);
    };
    std::integral_constant<int, 0> OPA;
    std::integral_constant<int, 1> OPW;

    f32x16 acc[NI][MI];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][i][r] = 0.f;

    int a_addr[4], w_addr[4];          // this lane's fragment of k-step kk, block 0 of the wave's rows (A) / columns (W), buffer 0; block i: + i * 4096
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_addr[kk] = lds0 + lds_tile_off(wm * 128 + l31, 2 * kk + hi);
        w_addr[kk] = lds0 + 2 * E_HALF + lds_tile_off(wn * 128 + l31, 2 * kk + hi);
    }
    u32x4 fa[2][MI], fw[2][NI];        // fragment registers: set kk & 1 holds k-step kk
    // one k-step: 16 MFMAs on fragment set CUR, between them the 8 LDS reads of the next k-step into the other set (from `rbuf` — buffer offset 0 / E_BUF —
    // at k-step index RK) and, where DMAOP >= 0, the wave's 8 pieces of that operand of K tile dkt into buffer db
    auto kstep = [&](auto curc, auto rkc, int rbuf, auto readc, auto dmaopc, int dkt, int db) __attribute__((always_inline)) {
        constexpr bool do_read = decltype(readc)::value;
        constexpr int CUR = decltype(curc)::value, RK = decltype(rkc)::value, DMAOP = decltype(dmaopc)::value;
        constexpr int NXT = CUR ^ 1;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int j = n * MI + i;
                acc[n][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[CUR][n]), __builtin_bit_cast(bf16x8, fa[CUR][i]), acc[n][i], 0, 0, 0);
                if constexpr (do_read) {
                    // reads in the order the next k-step consumes them: W(0), A(0..3), W(1..3)
                    if (j == 0) fw[NXT][0] = *(lds_u32x4_t)(w_addr[RK] + rbuf);
                    else if (j >= 1 && j <= 4) fa[NXT][j - 1] = *(lds_u32x4_t)(a_addr[RK] + rbuf + (j - 1) * 4096);
                    else if (j >= 5 && j <= 7) fw[NXT][j - 4] = *(lds_u32x4_t)(w_addr[RK] + rbuf + (j - 4) * 4096);
                }
#ifndef SVI_W4_NODMA
                if constexpr (DMAOP >= 0) {
#if defined(SVI_W4_SPREAD)
                    if (j & 1) dma(std::integral_constant<int, DMAOP < 0 ? 0 : DMAOP>{}, j >> 1, dkt, db);
#elif defined(SVI_W4_EARLY)
                    if (j < 8) dma(std::integral_constant<int, DMAOP < 0 ? 0 : DMAOP>{}, j, dkt, db);
#else
                    if (j >= 8) dma(std::integral_constant<int, DMAOP < 0 ? 0 : DMAOP>{}, j - 8, dkt, db);
#endif
                }
#endif
                W4_KSTEP_FENCE();
            }
        }
    };
    std::integral_constant<int, 0> C0, K0;
    std::integral_constant<int, 1> C1, K1;
    std::integral_constant<int, 2> K2;
    std::integral_constant<int, 3> K3;
    std::integral_constant<int, -1> NODMA;

    std::true_type YES;
    std::false_type NO;
    // one K tile.  H1: tile t+1 exists (its W is staged here, its first fragments are read here);  H2: tile t+2 exists (its A is staged here)
    auto ktile = [&](int t, auto h1c, auto h2c) __attribute__((always_inline)) {
        constexpr bool H1 = decltype(h1c)::value, H2 = decltype(h2c)::value;
        const int b = t & 1;
        const int cur = b * E_BUF, oth = (b ^ 1) * E_BUF;
        if constexpr (H1) kstep(C0, K1, cur, YES, OPW, t + 1, b ^ 1);      // W of tile t+1 -> the other buffer (its A went there one tile ago / in the prologue)
        else kstep(C0, K1, cur, YES, NODMA, 0, 0);
        kstep(C1, K2, cur, YES, NODMA, 0, 0);
        kstep(C0, K3, cur, YES, NODMA, 0, 0);
        E_WAIT("s_waitcnt vmcnt(0) lgkmcnt(0)");          // this wave's reads of buffer b are complete; its pieces of tile t+1 have landed
        E_BAR();
        if constexpr (H2) kstep(C1, K0, oth, YES, OPA, t + 2, b);
        else if constexpr (H1) kstep(C1, K0, oth, YES, NODMA, 0, 0);
        else kstep(C1, K0, oth, NO, NODMA, 0, 0);
    };

    // prologue: tile 0 whole, the A tile of tile 1 behind it (W of tile 1 is staged by tile 0's first k-step)
    for (int j = 0; j < 8; ++j) dma(OPA, j, 0, 0);
    for (int j = 0; j < 8; ++j) dma(OPW, j, 0, 0);
    if (nk > 1) {
        for (int j = 0; j < 8; ++j) dma(OPA, j, 1, 1);
        E_WAIT("s_waitcnt vmcnt(8)");          // tile 0 has landed (this wave's pieces); A of tile 1 may still fly
    } else {
        E_WAIT("s_waitcnt vmcnt(0)");
    }
    E_BAR();
    fw[0][0] = *(lds_u32x4_t)(w_addr[0]);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[0][i] = *(lds_u32x4_t)(a_addr[0] + i * 4096);
#pragma unroll
    for (int n = 1; n < NI; ++n) fw[0][n] = *(lds_u32x4_t)(w_addr[0] + n * 4096);
    W4_KSTEP_FENCE();
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(t, YES, YES);
    if (nk >= 2) { ktile(t, YES, NO); ++t; }
    ktile(t, NO, NO);
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("s_nop 7" : "+v"(acc[n][i]));      // MFMA result -> VALU read
    __syncthreads();                                       // every wave is done with the operand buffers: the C tile may overwrite them
    gemm256_epilogue<MI, NI, TN, false, 256>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, 0);
}

// -------------------------------------------------------------------------------------------------
// "w4r": the four-wave tile fed THROUGH REGISTERS.  What the w4 kernel above showed (tools/gemm_ab.py, profiles/r6e_*): with its global loads taken out the
// loop runs 1504 TFLOP/s on the K = 8960 shape (the eight-wave kernel: 1344; hipBLASLt: 1485) — with them 1200: a `buffer_load ... lds` costs its wave ~60
// cycles of issue (MI355X_MICROARCH.md), a lone wave has no partner to multiply meanwhile, and sixteen pieces per K tile leave the matrix pipe idle a fifth of
// the time wherever they are placed.  Here the operands take the detour the vendor kernels take: plain buffer_load_dwordx4 into registers (two sets of 64 —
// a wave's 16 KiB share of a K tile — beside 256 accumulators and 64 fragment registers), written to LDS with ds_write_b128 one tile later.  The staging
// registers are a third and fourth pipeline stage: tile t+3 is requested while tile t multiplies (a window of ~3000 cycles instead of ~1000), and no wait at
// the barrier concerns vector memory at all.  Same LDS image (the source-side swizzle: lane-linear 16-byte writes), same MFMA order: the same bits.
//   tile t (parity P = t & 1; LDS buffer P holds it; register set P holds tile t+2, requested during tile t-1):
//   k-step 0   read k-step 1          request A of tile t+3 -> set P^1 (8 loads, one per MFMA of the second half)
//   k-step 1   read k-step 2          request W of tile t+3 -> set P^1
//   k-step 2   read k-step 3
//   s_waitcnt lgkmcnt(0) (this wave's reads of buffer P and its writes of tile t+1 into buffer P^1 are complete), s_barrier
//   k-step 3   read k-step 0 of tile t+1 (buffer P^1)          write set P (tile t+2) -> buffer P, one ds_write_b128 per MFMA
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void gemm_bf16_nt_w4r_kernel(SviGemmArgs g, int tiles_m, int tiles_n, int GM) {
    constexpr int MI = 4, NI = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lds0 = (int)(size_t)(lptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
    const int group = swz / (GM * tiles_n);
    const int first_m = group * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int in_group = swz - group * GM * tiles_n;
    const int tile_n = in_group / gm;
    const int tile_m = first_m + (in_group - tile_n * gm);
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    if (g.W2 && n0 >= g.n_split) {
        g.W = g.W2 - (size_t)g.n_split * g.ldw;
        g.bias = g.bias2 ? g.bias2 - g.n_split : nullptr;
    }
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.A), 0, (int)(((unsigned)(g.M - 1) * (unsigned)g.lda + (unsigned)g.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(g.W), 0, (int)(((unsigned)(g.N - 1) * (unsigned)g.ldw + (unsigned)g.K) * 2u), 0x00020000);
    const int r8 = wave * 8 + (lane >> 3);
    const int c8 = (lane & 7) ^ ((r8 >> 1) & 7);
    const int a_vo = (r8 * g.lda + c8 * 8) * 2, w_vo = (r8 * g.ldw + c8 * 8) * 2;
    const unsigned a_so0 = (unsigned)m0 * (unsigned)g.lda * 2u, w_so0 = (unsigned)n0 * (unsigned)g.ldw * 2u;
    const unsigned a_j = 32u * (unsigned)g.lda * 2u, w_j = 32u * (unsigned)g.ldw * 2u;
    const int nk = g.K / BK;
    const int st_addr = lds0 + wave * 1024 + lane * 16;          // this lane's 16 bytes of piece 0 of the A tile of buffer 0 (piece j: + j * 4096; W: + 2 * E_HALF)
    u32x4 sa[2][8], sw[2][8];                                    // the two staging sets: a wave's 8 + 8 pieces of one K tile each
    auto gload = [&](auto opc, auto setc, int j, int kt) __attribute__((always_inline)) {
        constexpr int op = decltype(opc)::value, S = decltype(setc)::value;
        // a K tile past the last one is requested at an offset beyond the descriptor's range: the request returns zeros without touching memory, so the
        // loop's last tiles run the same instruction stream as every other (what they stage is never read)
        const unsigned kb = kt < nk ? (unsigned)kt * (BK * 2) : 0x80000000u;
        if constexpr (op == 0) sa[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_vo, (int)(a_so0 + kb + (unsigned)j * a_j), 0);
        else sw[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_vo, (int)(w_so0 + kb + (unsigned)j * w_j), 0);
    };
    typedef __attribute__((address_space(3))) u32x4* lds_w_u32x4_t;
    auto swrite = [&](auto opc, auto setc, int j, int boff) __attribute__((always_inline)) {
        constexpr int op = decltype(opc)::value, S = decltype(setc)::value;
        if constexpr (op == 0) *(lds_w_u32x4_t)(st_addr + boff + j * 4096) = sa[S][j];
        else *(lds_w_u32x4_t)(st_addr + boff + 2 * E_HALF + j * 4096) = sw[S][j];
    };
    std::integral_constant<int, 0> OPA, S0;
    std::integral_constant<int, 1> OPW, S1;

    f32x16 acc[NI][MI];
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][i][r] = 0.f;
    int a_addr[4], w_addr[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_addr[kk] = lds0 + lds_tile_off(wm * 128 + l31, 2 * kk + hi);
        w_addr[kk] = lds0 + 2 * E_HALF + lds_tile_off(wn * 128 + l31, 2 * kk + hi);
    }
    u32x4 fa[2][MI], fw[2][NI];
    // one k-step: 16 MFMAs on fragment set CUR; MFMAs 0..7 are each followed by one LDS read of the next k-step (set CUR^1, from buffer offset rbuf at k-step
    // RK).  MODE: 0 nothing more; 1 / 2: MFMAs 8..15 each request one A / W piece of K tile xkt into staging set SET; 3: every MFMA is followed by one
    // ds_write_b128 of staging set SET (A pieces behind MFMAs 0..7, W pieces behind 8..15) into the buffer at offset xoff
    auto kstep = [&](auto curc, auto rkc, int rbuf, auto readc, auto modec, auto setc, int xkt, int xoff) __attribute__((always_inline)) {
        constexpr int CUR = decltype(curc)::value, RK = decltype(rkc)::value, MODE = decltype(modec)::value;
        constexpr bool do_read = decltype(readc)::value;
        constexpr int NXT = CUR ^ 1;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int j = n * MI + i;
                acc[n][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[CUR][n]), __builtin_bit_cast(bf16x8, fa[CUR][i]), acc[n][i], 0, 0, 0);
                if constexpr (do_read) {
                    if (j == 0) fw[NXT][0] = *(lds_u32x4_t)(w_addr[RK] + rbuf);
                    else if (j >= 1 && j <= 4) fa[NXT][j - 1] = *(lds_u32x4_t)(a_addr[RK] + rbuf + (j - 1) * 4096);
                    else if (j >= 5 && j <= 7) fw[NXT][j - 4] = *(lds_u32x4_t)(w_addr[RK] + rbuf + (j - 4) * 4096);
                }
                if constexpr (MODE == 1) { if (j >= 8) gload(OPA, setc, j - 8, xkt); }
                if constexpr (MODE == 2) { if (j >= 8) gload(OPW, setc, j - 8, xkt); }
                if constexpr (MODE == 3) {
                    if (j < 8) swrite(OPA, setc, j, xoff);
                    else swrite(OPW, setc, j - 8, xoff);
                }
                W4_KSTEP_FENCE();
            }
        }
    };
    std::integral_constant<int, 0> C0, K0, M0_;
    std::integral_constant<int, 1> C1, K1, M1_;
    std::integral_constant<int, 2> K2, M2_;
    std::integral_constant<int, 3> K3, M3_;
    std::true_type YES;
    std::false_type NO;
    // one K tile of parity P (every tile runs this one stream: requests for tiles past the end return zeros, what is written or read for them is never used)
    auto ktile = [&](int t, auto pc) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
        std::integral_constant<int, P> SP;
        std::integral_constant<int, P ^ 1> SQ;
        constexpr int cur = P * E_BUF, oth = (P ^ 1) * E_BUF;
        kstep(C0, K1, cur, YES, M1_, SQ, t + 3, 0);
        kstep(C1, K2, cur, YES, M2_, SQ, t + 3, 0);
        kstep(C0, K3, cur, YES, M0_, SQ, 0, 0);
        E_WAIT("s_waitcnt lgkmcnt(0)");          // this wave's reads of buffer P are complete, and its writes of tile t+1 into buffer P^1 (previous tile) have landed
        E_BAR();
        kstep(C1, K0, oth, YES, M3_, SP, 0, cur);
    };

    // prologue: tiles 0 and 1 into LDS through the staging sets, tile 2 requested into set 0 (tile 3 is requested by tile 0's first k-steps into set 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) { gload(OPA, S0, j, 0); gload(OPW, S0, j, 0); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { gload(OPA, S1, j, 1); gload(OPW, S1, j, 1); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { swrite(OPA, S0, j, 0); swrite(OPW, S0, j, 0); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { gload(OPA, S0, j, 2); gload(OPW, S0, j, 2); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { swrite(OPA, S1, j, E_BUF); swrite(OPW, S1, j, E_BUF); }
    E_WAIT("s_waitcnt lgkmcnt(0)");
    E_BAR();
    fw[0][0] = *(lds_u32x4_t)(w_addr[0]);
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[0][i] = *(lds_u32x4_t)(a_addr[0] + i * 4096);
#pragma unroll
    for (int n = 1; n < NI; ++n) fw[0][n] = *(lds_u32x4_t)(w_addr[0] + n * 4096);
    W4_KSTEP_FENCE();
    for (int t = 0; t < nk; t += 2) {                      // both parities per trip: the staging sets and LDS buffers are named at compile time
        ktile(t, S0);
        ktile(t + 1, S1);                                   // (the launcher sends an odd number of K tiles to the eight-wave kernel)
    }
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("s_nop 7" : "+v"(acc[n][i]));
    __syncthreads();
    gemm256_epilogue<MI, NI, TN, false, 256>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi, 0);
}

#endif  // SVI_GEMM_EXPERIMENTS

// -------------------------------------------------------------------------------------------------
// Skinny-M kernel (SviGemmArgs.skinny, M <= 128): the text encoder's projections have 10^1..10^2 rows against 4096..10240 columns —
// the job is to stream the WEIGHTS once at HBM speed, and a 128^2 tiling offers the chip 32..80 workgroups.  Here a workgroup owns
// 16 output columns for all rows, its waves split K, fragments go global -> register -> MFMA (v_mfma_f32_16x16x32_bf16, no
// LDS staging: every weight element is used once), the partial sums meet in LDS.  N / 16 workgroups (256 for N = 4096) of eight
// waves, 16 B per lane per load, eight K steps in flight per wave (64 KiB of weights per CU).  The activations (<= 128 x K) are re-read by every workgroup from L2.
// Operand roles as in the other kernels: the weight fragment is the first MFMA operand, so a lane ends up holding 4 consecutive
// columns of one row -> 8-byte stores.  Epilogues: bias, and bias + residual with the module output rounded first.
// -------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4s;
// MB: 16-row blocks (M <= 16 MB); NW: waves = K splits; UN: K steps requested together (bytes in flight per wave = UN KiB of weights)
template <int MB, int NW, int UN>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(SviGemmArgs g) {
    __shared__ f32x4s red[NW][MB][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kq = g.K / NW, kbeg = wave * kq, kend = kbeg + kq;          // K % (32 NW) == 0: every share is a whole number of 32-steps
    const bf16* wp = g.W + (size_t)min(n0 + r16, g.N - 1) * g.ldw + 8 * kg;
    const bf16* ap[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) ap[b] = g.A + (size_t)min(16 * b + r16, g.M - 1) * g.lda + 8 * kg;
    f32x4s acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[b] = f32x4s{0.f, 0.f, 0.f, 0.f};
    int k = kbeg;
    for (; k + 32 * UN <= kend; k += 32 * UN) {                  // UN K steps requested together, then consumed
        bf16x8 wf[UN], af[UN][MB];
#pragma unroll
        for (int u = 0; u < UN; ++u) wf[u] = ld_bf16x8(wp + k + 32 * u);          // the HBM stream first
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int b = 0; b < MB; ++b) af[u][b] = ld_bf16x8(ap[b] + k + 32 * u);
        __builtin_amdgcn_sched_barrier(0);                        // all requests out before the first wait (hipcc would interleave them with the MFMAs)
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int b = 0; b < MB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[u][b], acc[b], 0, 0, 0);
    }
    for (; k < kend; k += 32) {
        const bf16x8 wf = ld_bf16x8(wp + k);
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, ld_bf16x8(ap[b] + k), acc[b], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) red[wave][b][lane] = acc[b];
    __syncthreads();
    for (int idx = tid; idx < MB * 64; idx += 64 * NW) {
        const int b = idx >> 6, l = idx & 63;
        f32x4s s = red[0][b][l];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[w][b][l];
        const int m = 16 * b + (l & 15), n = n0 + 4 * (l >> 4);
        if (m >= g.M || n >= g.N) continue;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = s[e];
            if (n + e < g.N) {
                if (g.bias) y += (float)g.bias[n + e];
                y = rbf(y);
                if (g.epi == SVI_EPI_BIAS_GATE_RES) {
                    if (g.gate) y = rbf(g.gate[n + e] * y);
                    y += (float)g.res[(size_t)m * g.ldres + n + e];
                }
            }
            o[e] = (bf16)y;
        }
        if (n + 4 <= g.N) *reinterpret_cast<bf16x4*>(g.C + (size_t)m * g.ldc + n) = o;
        else
            for (int e = 0; e < 4 && n + e < g.N; ++e) g.C[(size_t)m * g.ldc + n + e] = o[e];
    }
}

// =================================================================================================
// MX-fp8 path (opt-in; never the default): C[M,N] = epi(A8[M,K] · W8[N,K]^T) on v_mfma_scale_f32_32x32x64_f8f6f4 — OCP e4m3
// elements with one E8M0 power-of-two scale per 32 consecutive K elements, dequantised inside the matrix pipe at twice the bf16
// rate (MI355X_MICROARCH.md: MX K=128 ~4.6 PFLOP/s).  north_star names "bf16/fp8 MFMA" GEMMs; the reference itself only STORES
// weights as e4m3 (test_svi.py:337, vram_management/layers.py:65-71) and computes in bf16, so this is arithmetic the reference never
// performs: its own tolerance, its own bench line, never the headline.
//   weights      the reference's e4m3 storage bytes as they are, unit scales (2^0): exactly the values its FP8 mode multiplies with
//   activations  quantised per row and 32-element K block (mx8_quantize_kernel): E = max(exponent(amax) - 8, 0) biased by 127
//                (OCP MX: shared scale 2^(floor(log2 amax) - emax), emax(e4m3) = 8), element = e4m3_rne_sat(x * 2^-(E-127))
//   scale layout [K / 128][rows_padded] dwords: byte b of dword [kt][m] scales K block 4 kt + b of row m — a k-tile's scales for a 256-row
//                tile are one contiguous KiB = ONE LDS-DMA piece
// Same 256 x 256 tile, LDS image (128-byte rows, source-side XOR swizzle), XCD band order and epilogue as the bf16 kernels; a K tile is
// 128 elements = the same 128 bytes per row, fed to 2 x 8 MFMAs of 64 cycles instead of 4 x 8 of 32.
// Operand layout of the scaled MFMA, measured (tools/mx8_layout_probe.py) and pinned by tests/test_gpu_mx8.py: within a 64-wide K step lane l
// (row l & 31, half hi = l >> 5) supplies K elements [16 hi, +16) in its first four VGPRs and [32 + 16 hi, +16) in its last four — two
// K = 32 halves, each spread over both lane halves — while the E8M0 scale of the FIRST 32-element block is taken from the hi = 0 lanes and that
// of the SECOND block from the hi = 1 lanes (the byte of the scale VGPR that op_sel names).  So a lane's two 16-byte LDS reads are chunks
// 4 s + hi and 4 s + 2 + hi of the 128-byte row, and lane half hi carries the scale of block 2 s + hi.
// =================================================================================================
typedef __attribute__((ext_vector_type(8))) int i32x8;
#define MX8_SC_OFF (4 * T_STAGE)                 // two 1 KiB scale slabs behind the four operand stages
static_assert(MX8_SC_OFF + 2048 <= LDS256_BYTES, "scale slabs must fit the 256^2 kernels' LDS allocation");

struct SviMx8Args {
    SviGemmArgs g;                 // A = e4m3 activations [M, lda bytes], W = e4m3 weights [N, ldw bytes]; C / bias / epilogue as for bf16
    const unsigned* a_scales;      // [K / 128][sc_rows]: block scales of the SCALED operand's rows (A's rows; with WSC the W operand's rows)
    int sc_rows;
};

// x bf16 [rows, ldx] -> q e4m3 [rows, ldq] + scales [K/128][sc_rows].  One lane per 8 elements, four lanes per 32-block, sixteen per dword of scales.
__global__ __launch_bounds__(256) void mx8_quantize_kernel(const bf16* __restrict__ x, int ldx, int rows, int K, unsigned char* __restrict__ q, int ldq,
                                                           unsigned* __restrict__ scales, int sc_rows) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int per_row = K >> 3;
    const long total = (long)rows * per_row;
    const bool live = idx < total;
    const int row = live ? (int)(idx / per_row) : 0, c8 = live ? (int)(idx - (long)row * per_row) : 0;
    float v[8];
    if (live) {
        const bf16x8 t = ld_bf16x8(x + (size_t)row * ldx + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    mx8_quant8(v, live, row, c8, q, ldq, scales, sc_rows);
}

__device__ __forceinline__ i32x8 mx8_frag(int base, int s, int hi, int row) {      // K bytes [16 hi, +16) and [32 + 16 hi, +16) of step s of `row`
    const u32x4 lo = *(lds_u32x4_t)(base + lds_tile_off(row, 4 * s + hi));
    const u32x4 up = *(lds_u32x4_t)(base + lds_tile_off(row, 4 * s + 2 + hi));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)up[0], (int)up[1], (int)up[2], (int)up[3]};
}

// WSC (round 6): the block scales belong to the W operand's rows and the A operand carries unit scales — the transposed projection V^T = Wv · X^T, where the
// stored e4m3 weight is the A operand (output rows = channels) and the quantised activation the W operand (output columns = tokens).
template <bool WSC>
__global__ __launch_bounds__(512, 2) void gemm_mx8_nt_256_kernel(SviMx8Args a, int tiles_m, int tiles_n, int GM) {
    const SviGemmArgs& g = a.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lds0 = (int)(size_t)(lptr_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;

    const int nwg = tiles_m * tiles_n;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
    const int group = swz / (GM * tiles_n);
    const int first_m = group * GM;
    const int gm = min(GM, tiles_m - first_m);
    const int in_group = swz - group * GM * tiles_n;
    const int tile_n = in_group / gm;
    const int tile_m = first_m + (in_group - tile_n * gm);
    const int m0 = tile_m * TM, n0 = tile_n * TN;

    const unsigned char* A8 = reinterpret_cast<const unsigned char*>(g.A);
    const unsigned char* W8 = reinterpret_cast<const unsigned char*>(g.W);
    unsigned a_off[4], w_off[4];         // byte offsets of this lane's source chunk at k = 0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        a_off[j] = (unsigned)min(m0 + r, g.M - 1) * (unsigned)g.lda + c * 16;
        w_off[j] = (unsigned)min(n0 + r, g.N - 1) * (unsigned)g.ldw + c * 16;
    }
    const int nk = g.K / 128;
    auto stage = [&](int kt, int buf) {
        char* As = smem + buf * 2 * T_STAGE;
        char* Ws = As + T_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(A8 + (size_t)kt * 128 + a_off[j]), (lptr_t)(As + (j * 8 + wave) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(W8 + (size_t)kt * 128 + w_off[j]), (lptr_t)(Ws + (j * 8 + wave) * 1024), 16, 0, 0);
        }
        if (wave == 0)        // the k-tile's scale dwords of rows m0 .. m0+255 (WSC: of the W operand's rows n0 .. n0+255): one KiB, lane-linear (the scale table is padded to whole tiles)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.a_scales + (size_t)kt * a.sc_rows + (WSC ? n0 : m0) + lane * 4), (lptr_t)(smem + MX8_SC_OFF + buf * 1024), 16, 0, 0);
    };

    f32x16 acc[2][4];                       // [ni][mi]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const int abase = lds0 + cur * 2 * T_STAGE, wbase = abase + T_STAGE, sbase = lds0 + MX8_SC_OFF + cur * 1024;
        int xs[4];          // WSC: xs[0..1] are the scales of the wave's two W row blocks
#pragma unroll
        for (int mi = 0; mi < (WSC ? 2 : 4); ++mi)
            xs[mi] = (int)((*(const __attribute__((address_space(3))) unsigned*)(sbase + ((WSC ? wn * 64 : wm * 128) + mi * 32 + l31) * 4)) >> (8 * hi));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            i32x8 xa[4], wb[2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) xa[mi] = mx8_frag(abase, s, hi, wm * 128 + mi * 32 + l31);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) wb[ni] = mx8_frag(wbase, s, hi, wn * 64 + ni * 32 + l31);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (WSC) {
                        if (s == 0) acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0, xs[ni], 0, 0x7f7f7f7f);
                        else acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[ni], xa[mi], acc[ni][mi], 0, 0, 2, xs[ni], 0, 0x7f7f7f7f);
                    } else {
                        if (s == 0) acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0, 0x7f7f7f7f, 0, xs[mi]);
                        else acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0, 0x7f7f7f7f, 2, xs[mi]);
                    }
                }
        }
        __syncthreads();                    // drains the LDS-DMA of tile kt+1 (vmcnt(0)) and fences the reads of tile kt
    }
    if (g.q8) gemm256_epilogue<4, 2, TN, true>(g, acc, smem, m0, n0, tid, wm, wn, l31, hi);      // the result leaves as MX e4m3 + block scales (the next GEMM's operand)
    else gemm256_epilogue(g, acc, smem, m0, n0, tid, wm, wn, l31, hi);
}

svi_status svi_launch_mx8_quantize(const bf16* x, int ldx, int rows, int K, unsigned char* q, int ldq, unsigned* scales, int sc_rows, hipStream_t st) {
    SVI_REQUIRE(rows > 0 && K > 0 && K % 128 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && ldx >= K && ldq >= K, "mx8 quantize: K=%d must be a multiple of 128 (ldx %d, ldq %d)", K, ldx, ldq);
    SVI_REQUIRE(sc_rows >= rows && ((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 8) == 0, "mx8 quantize: bad scale rows / alignment");
    const long total = (long)rows * (K / 8);
    hipLaunchKernelGGL(mx8_quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, ldx, rows, K, q, ldq, scales, sc_rows);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// sc_rows: row count of the scale table, a multiple of 256 (whole tiles: the scale slab of a tile is fetched unconditionally)
svi_status svi_launch_gemm_mx8(const SviGemmArgs& g, const unsigned* a_scales, int sc_rows, hipStream_t st) {
    SVI_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 128 == 0, "mx8 gemm: K=%d must be a multiple of 128", g.K);
    SVI_REQUIRE(g.lda % 16 == 0 && g.ldw % 16 == 0 && g.ldc % 8 == 0 && g.lda >= g.K && g.ldw >= g.K && g.ldc >= g.N, "mx8 gemm: bad leading dims");
    SVI_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)a_scales % 16) == 0, "mx8 gemm: operands must be 16-byte aligned");
    SVI_REQUIRE(sc_rows % 256 == 0 && sc_rows >= ((g.M + 255) / 256) * 256, "mx8 gemm: the scale table must cover whole 256-row tiles (sc_rows %d, M %d)", sc_rows, g.M);
    SVI_REQUIRE(g.epi >= 0 && g.epi <= SVI_EPI_BIAS_RELU, "mx8 gemm: unknown epilogue %d", g.epi);
    SVI_REQUIRE((long)g.M * g.lda < (1L << 31) && (long)g.N * g.ldw < (1L << 31), "mx8 gemm: operand beyond 2 GiB");
    if (g.epi == SVI_EPI_BIAS_GATE_RES) SVI_REQUIRE(g.res != nullptr && g.ldres % 8 == 0 && ((uintptr_t)g.res % 16) == 0, "mx8 gemm: gate/residual epilogue needs an aligned residual");
    if (g.q8) SVI_REQUIRE(g.q8s && g.N % 256 == 0 && g.ldq8 >= g.N && g.ldq8 % 8 == 0 && ((uintptr_t)g.q8 % 8) == 0 && ((uintptr_t)g.q8s % 4) == 0 &&
                          g.q8_sc_rows >= g.M && (const void*)g.q8 != (const void*)g.A && (const void*)g.q8s != (const void*)a_scales,
                          "mx8 gemm: quantised output needs N %% 256 == 0 (N=%d), its own buffers and a scale table covering M", g.N);
    const int tm = (g.M + TM - 1) / TM, tn = (g.N + TN - 1) / TN;
    const int gm_rows = svi_switches().gemm_gm ? svi_switches().gemm_gm : (tn >= 16 ? 5 : 2);
    SviMx8Args a{g, a_scales, sc_rows};
    SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_mx8_nt_256_kernel<false>), LDS256_BYTES));
    hipLaunchKernelGGL(gemm_mx8_nt_256_kernel<false>, dim3(tm * tn), dim3(512), LDS256_BYTES, st, a, tm, tn, gm_rows);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// The same GEMM with the block scales on the W operand's rows (w_scales [K / 128][sc_rows], sc_rows a multiple of 256 covering N) and unit scales on A:
// C[M, N] = epi(A8[M, K] · dequant(W8[N, K])^T).  The DiT's transposed value projection: A8 = the stored e4m3 weight [dim, K], W8 = the quantised
// activation rows [tokens, K], C = V^T [dim, ldc >= tokens], bias along M.
svi_status svi_launch_gemm_mx8_wscaled(const SviGemmArgs& g, const unsigned* w_scales, int sc_rows, hipStream_t st) {
    SVI_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 128 == 0, "mx8 gemm (W scaled): K=%d must be a multiple of 128", g.K);
    SVI_REQUIRE(g.lda % 16 == 0 && g.ldw % 16 == 0 && g.ldc % 8 == 0 && g.lda >= g.K && g.ldw >= g.K && g.ldc >= g.N, "mx8 gemm (W scaled): bad leading dims");
    SVI_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0 && ((uintptr_t)g.C % 16) == 0 && ((uintptr_t)w_scales % 16) == 0, "mx8 gemm (W scaled): operands must be 16-byte aligned");
    SVI_REQUIRE(sc_rows % 256 == 0 && sc_rows >= ((g.N + 255) / 256) * 256, "mx8 gemm (W scaled): the scale table must cover whole 256-row tiles of the W operand (sc_rows %d, N %d)", sc_rows, g.N);
    SVI_REQUIRE(g.epi == SVI_EPI_BIAS && !g.q8 && !g.rowss && !g.W2, "mx8 gemm (W scaled): the plain bias epilogue only");
    SVI_REQUIRE((long)g.M * g.lda < (1L << 31) && (long)g.N * g.ldw < (1L << 31), "mx8 gemm (W scaled): operand beyond 2 GiB");
    const int tm = (g.M + TM - 1) / TM, tn = (g.N + TN - 1) / TN;
    const int gm_rows = svi_switches().gemm_gm ? svi_switches().gemm_gm : (tn >= 16 ? 5 : 2);
    SviMx8Args a{g, w_scales, sc_rows};
    SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_mx8_nt_256_kernel<true>), LDS256_BYTES));
    hipLaunchKernelGGL(gemm_mx8_nt_256_kernel<true>, dim3(tm * tn), dim3(512), LDS256_BYTES, st, a, tm, tn, gm_rows);
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}

// Which kernel a bf16 GEMM of this shape runs on — pure arithmetic on sizes, switches and the device's CU count (svi_gemm_plan exposes it;
// tests/test_plans.py): 0 = weight-streaming skinny kernel, 128 = 128^2 tile, 192 = 256 x 192 tile, 259 / 260 = the 256^2 tile with four / two phases per K tile.
int svi_gemm_choose(const SviGemmArgs& g, int ncu) {
    if (g.skinny && g.M <= 128 && g.K % 128 == 0 && !g.bias_along_m && (g.epi == SVI_EPI_BIAS || g.epi == SVI_EPI_BIAS_GATE_RES) && g.ldc % 4 == 0) return 0;
    if (ncu <= 0) ncu = 256;
    const int selM = g.sel_m > 0 ? g.sel_m : g.M, selN = g.sel_n > 0 ? g.sel_n : g.N;
    const long t256 = (long)((selM + TM - 1) / TM) * ((selN + TN - 1) / TN);
    // The 256-row kernels reach A and W through buffer descriptors with 32-bit byte offsets: an operand of 2^31 elements (4 GiB) or more takes the 128^2
    // kernel, whose loads use 64-bit pointers; C, the residual and the bias are addressed with 64-bit pointers in every kernel (tests/test_plans.py).
    const bool fits32 = (long)g.M * g.lda < (1L << 31) && (long)g.N * g.ldw < (1L << 31);
    const SviSwitches& sw = svi_switches();
    const bool want256 = sw.gemm_kernel ? sw.gemm_kernel >= 192 : (t256 >= ncu / 2);      // 256-row tiles when the problem fills at least half the chip with them
    if (!(want256 && g.K % BK == 0 && fits32)) return 128;
    // 256 x 192 tiles where they fill the chip's rounds better (sequence-parallel shards; see the kernel's header).  A 192-wide tile is
    // 0.75 of a 256-wide one plus ~8 % (6 MFMAs per 5 fragment reads instead of 8 per 6, the same DMA / barrier count per K tile; measured,
    // tools/gemm_ab.py shards: N = 1536 shard GEMMs gain 4-20 %, the N = 8960 ones — already many rounds of 256-wide tiles — lose 2-10 % and stay)
    const int tm_s = (selM + TM - 1) / TM;
    const long r256 = ((long)tm_s * ((selN + TN - 1) / TN) + ncu - 1) / ncu, r192 = ((long)tm_s * ((selN + 192 - 1) / 192) + ncu - 1) / ncu;
    const bool better = (double)r192 * 0.75 * 1.08 < (double)r256 * 0.97;
    if (g.N >= 192 && (sw.gemm_kernel == 192 || (sw.gemm_kernel == 0 && better))) return 192;
#ifdef SVI_GEMM_EXPERIMENTS
    if (sw.gemm_kernel == 265 && ((g.K / BK) & 1)) return SVI_GEMM_DEFAULT_256;          // the register-fed four-wave loop walks K tiles in pairs
    if (sw.gemm_kernel == 264 || sw.gemm_kernel == 265) return sw.gemm_kernel;
#endif
    if (sw.gemm_kernel == 259 || sw.gemm_kernel == 260) return sw.gemm_kernel;
    return SVI_GEMM_DEFAULT_256;
}

// compute units of the current device, asked once per device (the round arithmetic above; svi_launch_flash keeps the same kind of cache)
static int gemm_device_cus() {
    static std::atomic<int> cus[64];
    const int dev = svi_current_device();
    if (dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

svi_status svi_launch_gemm(const SviGemmArgs& g, hipStream_t st) {
    SVI_REQUIRE(g.M >= 0 && g.N >= 0 && g.K > 0, "gemm: bad sizes M=%d N=%d K=%d", g.M, g.N, g.K);
    if (g.M == 0 || g.N == 0) return SVI_OK;
    SVI_REQUIRE(g.K % 8 == 0, "gemm: K=%d must be a multiple of 8", g.K);
    SVI_REQUIRE(g.lda % 8 == 0 && g.ldw % 8 == 0 && g.ldc % 8 == 0, "gemm: leading dims must be multiples of 8");
    SVI_REQUIRE(g.lda >= g.K && g.ldw >= g.K && g.ldc >= g.N, "gemm: leading dims too small");
    SVI_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.W % 16) == 0 && ((uintptr_t)g.C % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    SVI_REQUIRE(g.epi >= 0 && g.epi <= SVI_EPI_BIAS_RELU, "gemm: unknown epilogue %d", g.epi);
    if (g.epi == SVI_EPI_BIAS_GATE_RES) {
        SVI_REQUIRE(g.res != nullptr && g.ldres % 8 == 0 && ((uintptr_t)g.res % 16) == 0,
                    "gemm: gate/residual epilogue needs an aligned residual");
    }
    const int kind = svi_gemm_choose(g, gemm_device_cus());
    const SviSwitches& sw = svi_switches();
    if (g.rowss)
        SVI_REQUIRE(kind != 0 && g.N % 64 == 0 && g.epi == SVI_EPI_BIAS && !g.bias_along_m && !g.W2 && g.ldss >= g.M && ((uintptr_t)g.rowss % 4) == 0,
                    "gemm: row statistics need a tiled kernel, N %% 64 == 0 (N = %d), the plain bias epilogue and ldss >= M", g.N);
    if (g.W2) {        // two weight matrices side by side: one launch when the split falls on a tile boundary of the kernel that runs, else two
        SVI_REQUIRE(g.n_split > 0 && g.n_split < g.N && !g.bias_along_m && ((uintptr_t)g.W2 % 16) == 0, "gemm: bad weight pair (n_split %d of N %d)", g.n_split, g.N);
        // tiles of the second half read rows n - n_split of W2 through a base moved back by n_split rows and a descriptor sized for N rows from there: the
        // second matrix must hold N - n_split rows of the same stride (the DiT's q | k: two [D, D] weights), and its own 2^31-element bound holds with A / W's
        SVI_REQUIRE((long)(g.N - g.n_split) * g.ldw < (1L << 31), "gemm: second weight matrix of 2^31 elements or more");
        const int tw = kind == 0 ? 0 : kind == 128 ? BN : kind == 192 ? 192 : TN;
        if (tw == 0 || g.n_split % tw != 0) {
            SviGemmArgs a = g, b = g;
            a.W2 = nullptr; a.bias2 = nullptr; a.n_split = 0; a.N = g.n_split; a.sel_n = g.sel_n > 0 ? g.sel_n : g.n_split;      // (the per-projection shape: the same kernel, and bits, as two separate launches)
            b.W2 = nullptr; b.bias2 = nullptr; b.n_split = 0; b.W = g.W2; b.bias = g.bias2; b.N = g.N - g.n_split; b.C = g.C + g.n_split;
            b.res = g.res ? g.res + g.n_split : nullptr; b.gate = g.gate ? g.gate + g.n_split : nullptr; b.sel_n = a.sel_n;
            SVI_TRY(svi_launch_gemm(a, st));
            return svi_launch_gemm(b, st);
        }
    }
    if (kind == 0) {
        dim3 grid((g.N + 15) / 16);
        const bool wide = g.K % 256 == 0;                 // eight K shares when they come out whole
#define SVI_SKINNY(MB, UN)                                                                                                        \
        do {                                                                                                                     \
            if (wide) hipLaunchKernelGGL((gemm_skinny_kernel<MB, 8, UN>), grid, dim3(512), 0, st, g);                              \
            else hipLaunchKernelGGL((gemm_skinny_kernel<MB, 4, UN>), grid, dim3(256), 0, st, g);                                   \
        } while (0)
        if (g.M <= 16) SVI_SKINNY(1, 8);
        else if (g.M <= 32) SVI_SKINNY(2, 8);
        else if (g.M <= 64) SVI_SKINNY(4, 8);
        else SVI_SKINNY(8, 4);
#undef SVI_SKINNY
        SVI_LAUNCH_CHECK();
        return SVI_OK;
    }
    if (kind != 128) {
        const int tm = (g.M + TM - 1) / TM, tn = (g.N + TN - 1) / TN;
        // measured (tools/gemm_gm.py on the staggered loop, round 4): N = 1536 (6 column panels): 4 is best (q/k/v 1143 vs 1112 TFLOP/s at 2, ffn2 1370 vs 1358);
        // N = 8960 (35 column panels, ffn1): 5 gives 1161 vs 1139 at 2 and 1093 at 8 — a 5 x 6 block of concurrent tiles per XCD
        // needs the fewest operand panels (HBM fetch per launch at 2: 2.2 GB against 0.13 GB of operands, profiles/r1h_gemm_ffn1_pmc.txt)
        const int gm_rows = sw.gemm_gm ? sw.gemm_gm : (tn >= 16 ? 5 : 4);
#ifdef SVI_ABLATIONS          // timing ablations (tools/gemm_epi_abl.py; results wrong when set): variant builds only
        const int abl = sw.gemm_epi_abl;
#else
        const int abl = 0;
#endif
        if (kind == 192) {            // the 256 x 192 tile (sequence-parallel shards)
            const int tn3 = (g.N + TN3 - 1) / TN3;
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_256e_kernel<2, 192>), LDS256_BYTES));
            hipLaunchKernelGGL((gemm_bf16_nt_256e_kernel<2, 192>), dim3(tm * tn3), dim3(512), LDS256_BYTES, st, g, tm, tn3, sw.gemm_gm ? sw.gemm_gm : (tn3 >= 16 ? 5 : 4), abl);
#ifdef SVI_GEMM_EXPERIMENTS
        } else if (kind == 265) {      // four waves, operands staged through registers
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_w4r_kernel), LDS256_BYTES));
            hipLaunchKernelGGL(gemm_bf16_nt_w4r_kernel, dim3(tm * tn), dim3(256), LDS256_BYTES, st, g, tm, tn, gm_rows);
        } else if (kind == 264) {      // four waves, 128 x 128 per wave
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_w4_kernel), LDS256_BYTES));
            hipLaunchKernelGGL(gemm_bf16_nt_w4_kernel, dim3(tm * tn), dim3(256), LDS256_BYTES, st, g, tm, tn, gm_rows);
#endif
        } else if (kind == 260) {
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_256e_kernel<2, 256>), LDS256_BYTES));
            hipLaunchKernelGGL((gemm_bf16_nt_256e_kernel<2, 256>), dim3(tm * tn), dim3(512), LDS256_BYTES, st, g, tm, tn, gm_rows, abl);
        } else {
            // Tried and dropped: starting the first round's workgroups out of phase (s_sleep by CU index) so that the CUs' store
            // bursts do not coincide: no gain on ffn1 (17.5 rounds), a loss wherever the tile count is a whole number of rounds.
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_256e_kernel<4, 256>), LDS256_BYTES));
            hipLaunchKernelGGL((gemm_bf16_nt_256e_kernel<4, 256>), dim3(tm * tn), dim3(512), LDS256_BYTES, st, g, tm, tn, gm_rows, abl);
        }
        SVI_LAUNCH_CHECK();
        return SVI_OK;
    }
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    // at most one workgroup per CU (the C1-size step's projections: 240 tiles): the loop with four K tiles of loads in flight; SVI_GEMM_PF = 1 | 4 forces one
    const bool deep = g.K % BK == 0 && g.K >= SVI_GEMM_DEEP_PF * BK && (sw.gemm_pf ? sw.gemm_pf > 1 : (long)tiles_m * tiles_n <= gemm_device_cus());
    if (deep) {
        if (sw.gemm_pf == 4) {          // (A/B: the four-wave form of the deep loop)
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<SVI_GEMM_DEEP_PF>), 4 * STAGE_BYTES));
            hipLaunchKernelGGL(gemm_bf16_nt_kernel<SVI_GEMM_DEEP_PF>, dim3(tiles_m * tiles_n), dim3(256), 4 * STAGE_BYTES, st, g, tiles_m, tiles_n);
        } else {
            SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>((gemm_bf16_nt_kernel<SVI_GEMM_DEEP_PF, 512>)), 4 * STAGE_BYTES));
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<SVI_GEMM_DEEP_PF, 512>), dim3(tiles_m * tiles_n), dim3(512), 4 * STAGE_BYTES, st, g, tiles_m, tiles_n);
        }
    } else {
        SVI_TRY(svi_ensure_lds(reinterpret_cast<const void*>(gemm_bf16_nt_kernel<1, 256>), 4 * STAGE_BYTES));
        hipLaunchKernelGGL((gemm_bf16_nt_kernel<1, 256>), dim3(tiles_m * tiles_n), dim3(256), 4 * STAGE_BYTES, st, g, tiles_m, tiles_n);
    }
    SVI_LAUNCH_CHECK();
    return SVI_OK;
}
