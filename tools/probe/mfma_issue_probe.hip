// mfma_issue_probe.hip — how many single-issue "filler" instructions hide in the shadow of a v_mfma_f32_32x32x16_bf16
// when ONE wave owns a SIMD (the regime of flash_fwd2_kernel)?  Each variant runs LOOPS iterations of 16 MFMAs with K
// fillers of one kind after every MFMA and reports cycles per MFMA (s_memtime over the loop, wave 0 of each block).
//   acc:  V = MFMA C/D in arch VGPRs, A = in AGPRs;  chains: number of independent accumulators rotated (1, 2, 4)
//   filler kinds: fma (v_fma_f32), exp (v_exp_f32), max3, cvt (v_cvt_pk_bf16_f32), nop (s_nop 0), ds (ds_read_b128 + wait later)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

#define FILL_fma "v_fma_f32 %[f0], %[f0], %[f1], %[f1]\n\t"
#define FILL_exp "v_exp_f32 %[f0], %[f1]\n\t"
#define FILL_max3 "v_max3_f32 %[f0], %[f0], %[f1], %[f2]\n\t"
#define FILL_cvt "v_cvt_pk_bf16_f32 %[f0], %[f1], %[f2]\n\t"
#define FILL_nop "s_nop 0\n\t"
#define FILL_add "v_add_f32 %[f0], %[f0], %[f1]\n\t"

#define F0(k)
#define F1(k) FILL_##k
#define F2(k) F1(k) F1(k)
#define F3(k) F2(k) F1(k)
#define F4(k) F2(k) F2(k)
#define F5(k) F4(k) F1(k)
#define F6(k) F4(k) F2(k)
#define F7(k) F4(k) F3(k)
#define F8(k) F4(k) F4(k)

// 4 MFMAs rotating over `chains` VGPR accumulators, each followed by the filler block
#define BODY_V(FILL)                                                                         \
    "v_mfma_f32_32x32x16_bf16 %[c0], %[a], %[b], %[c0]\n\t" FILL                            \
    "v_mfma_f32_32x32x16_bf16 %[c1], %[a], %[b], %[c1]\n\t" FILL                            \
    "v_mfma_f32_32x32x16_bf16 %[c2], %[a], %[b], %[c2]\n\t" FILL                            \
    "v_mfma_f32_32x32x16_bf16 %[c3], %[a], %[b], %[c3]\n\t" FILL
#define BODY_A(FILL)                                                                         \
    "v_mfma_f32_32x32x16_bf16 a[0:15], %[a], %[b], a[0:15]\n\t" FILL                        \
    "v_mfma_f32_32x32x16_bf16 a[16:31], %[a], %[b], a[16:31]\n\t" FILL                      \
    "v_mfma_f32_32x32x16_bf16 a[32:47], %[a], %[b], a[32:47]\n\t" FILL                      \
    "v_mfma_f32_32x32x16_bf16 a[48:63], %[a], %[b], a[48:63]\n\t" FILL

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, int loops, float seed) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed + 1; c2[r] = seed + 2; c3[r] = seed + 3; }
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float f0 = seed, f1 = seed * 0.5f, f2 = seed * 0.25f;
    asm volatile("; clobber" ::: "a0", "a15", "a16", "a31", "a32", "a47", "a48", "a63");
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < loops; ++i) {
#define RUN_V(FILL)                                                                                                   \
    asm volatile(REP4(BODY_V(FILL)) : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0)       \
                 : [a] "v"(a), [b] "v"(b), [f1] "v"(f1), [f2] "v"(f2));
#define RUN_V2(FILL) /* two chains only: c0, c1 alternate */                                                          \
    asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %[c0], %[a], %[b], %[c0]\n\t" FILL                                    \
                      "v_mfma_f32_32x32x16_bf16 %[c1], %[a], %[b], %[c1]\n\t" FILL)                                   \
                 : [c0] "+v"(c0), [c1] "+v"(c1), [f0] "+v"(f0) : [a] "v"(a), [b] "v"(b), [f1] "v"(f1), [f2] "v"(f2));
#define RUN_A(FILL)                                                                                                   \
    asm volatile(REP4(BODY_A(FILL)) : [f0] "+v"(f0) : [a] "v"(a), [b] "v"(b), [f1] "v"(f1), [f2] "v"(f2));
        if constexpr (VARIANT == 0) { RUN_V(F0(fma)) }
        else if constexpr (VARIANT == 1) { RUN_V(F1(fma)) }
        else if constexpr (VARIANT == 2) { RUN_V(F2(fma)) }
        else if constexpr (VARIANT == 3) { RUN_V(F3(fma)) }
        else if constexpr (VARIANT == 4) { RUN_V(F4(fma)) }
        else if constexpr (VARIANT == 5) { RUN_V(F5(fma)) }
        else if constexpr (VARIANT == 6) { RUN_V(F6(fma)) }
        else if constexpr (VARIANT == 7) { RUN_V(F7(fma)) }
        else if constexpr (VARIANT == 8) { RUN_V(F8(fma)) }
        else if constexpr (VARIANT == 10) { RUN_A(F0(fma)) }
        else if constexpr (VARIANT == 11) { RUN_A(F2(fma)) }
        else if constexpr (VARIANT == 12) { RUN_A(F4(fma)) }
        else if constexpr (VARIANT == 13) { RUN_A(F5(fma)) }
        else if constexpr (VARIANT == 14) { RUN_A(F6(fma)) }
        else if constexpr (VARIANT == 15) { RUN_A(F8(fma)) }
        else if constexpr (VARIANT == 20) { RUN_V2(F0(fma)) }
        else if constexpr (VARIANT == 21) { RUN_V2(F2(fma)) }
        else if constexpr (VARIANT == 22) { RUN_V2(F4(fma)) }
        else if constexpr (VARIANT == 23) { RUN_V2(F6(fma)) }
        else if constexpr (VARIANT == 30) { RUN_V(F1(exp)) }
        else if constexpr (VARIANT == 31) { RUN_V(F2(exp)) }
        else if constexpr (VARIANT == 32) { RUN_V(F3(exp)) }
        else if constexpr (VARIANT == 33) { RUN_V(F4(exp)) }
        else if constexpr (VARIANT == 34) { RUN_A(F2(exp)) }
        else if constexpr (VARIANT == 35) { RUN_A(F4(exp)) }
        else if constexpr (VARIANT == 40) { RUN_V(F2(max3)) }
        else if constexpr (VARIANT == 41) { RUN_V(F4(max3)) }
        else if constexpr (VARIANT == 42) { RUN_V(F2(cvt)) }
        else if constexpr (VARIANT == 43) { RUN_V(F4(cvt)) }
        else if constexpr (VARIANT == 44) { RUN_V(F4(nop)) }
        else if constexpr (VARIANT == 45) { RUN_V(F8(nop)) }
        else if constexpr (VARIANT == 46) { RUN_V(F4(add)) }
        // the B pair of flash_fwd2: exp exp add add cvt (5) / with two fma in front (7)
        else if constexpr (VARIANT == 50) { RUN_V(FILL_exp FILL_exp FILL_add FILL_add FILL_cvt) }
        else if constexpr (VARIANT == 51) { RUN_V(FILL_fma FILL_fma FILL_exp FILL_exp FILL_add FILL_add FILL_cvt) }
        else if constexpr (VARIANT == 52) { RUN_A(FILL_exp FILL_exp FILL_add FILL_add FILL_cvt) }
        else if constexpr (VARIANT == 53) { RUN_V2(FILL_exp FILL_exp FILL_add FILL_add FILL_cvt) }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = f0 + c0[0] + c1[1] + c2[2] + c3[3];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 123.456f) out[0] = 0;
}

extern "C" int probe_run(int variant, int loops, unsigned long long* host_out, float* ms_out) {
    unsigned long long* d = nullptr;
    const int blocks = 256;
    if (hipMalloc((void**)&d, blocks * 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
#define LAUNCH(V) case V: hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d, 8, 1.0f); hipDeviceSynchronize(); hipEventRecord(e0, 0); \
                          hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d, loops, 1.0f); hipEventRecord(e1, 0); break;
    switch (variant) {
        LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8)
        LAUNCH(10) LAUNCH(11) LAUNCH(12) LAUNCH(13) LAUNCH(14) LAUNCH(15)
        LAUNCH(20) LAUNCH(21) LAUNCH(22) LAUNCH(23)
        LAUNCH(30) LAUNCH(31) LAUNCH(32) LAUNCH(33) LAUNCH(34) LAUNCH(35)
        LAUNCH(40) LAUNCH(41) LAUNCH(42) LAUNCH(43) LAUNCH(44) LAUNCH(45) LAUNCH(46)
        LAUNCH(50) LAUNCH(51) LAUNCH(52) LAUNCH(53)
        default: return 2;
    }
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipEventElapsedTime(ms_out, e0, e1);
    hipMemcpy(host_out, d, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    hipFree(d);
    return 0;
}
