// What does a CU's global -> LDS fetch path deliver?   (round 6; DESIGN 4b "balanced on the fetch path")
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/probe/fetch_path_probe tools/probe/fetch_path_probe.hip && tools/probe/fetch_path_probe
// One 512-thread workgroup per CU (8 waves, as the 256-row GEMM), every wave streams 1 KiB pieces (64 lanes x 16 B, 8 rows x 128 B of a row-major
// matrix with a 17920-byte row stride = ffn2's A operand) out of a region that stays L2-resident, keeping WINDOW pieces in flight:
//   dma   buffer_load_dwordx4 ... lds   (what the GEMM / attention kernels use)
//   reg   buffer_load_dwordx4 into VGPRs (no LDS write)
// private: each CU walks its own 128 KiB of lines;  shared: the 32 CUs of an XCD walk the SAME lines (a W panel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WINDOW>
__global__ __launch_bounds__(512, 1) void fetch_kernel(const char* base, long region_stride, int rows, int row_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* reg = base + (long)blockIdx.x * region_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reg), 0, rows * row_bytes, 0x00020000);
    // a piece = 8 rows x 128 B; this wave's pieces: row blocks wave, wave + 8, ...; k advances 128 B per step
    const int vo = (lane >> 3) * row_bytes + (lane & 7) * 16;
    const int row_blocks = rows / 8, ksteps = row_bytes / 128;
    u32x4 acc = {0u, 0u, 0u, 0u};
    int issued = 0;
    for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < ksteps; ++k) {
#pragma unroll 4
            for (int rb = wave; rb < row_blocks; rb += 8) {
                const int so = rb * 8 * row_bytes + k * 128;
                if (MODE == 0) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + wave * 16384 + (issued & 15) * 1024), 16, vo, so, 0, 0);
                } else {
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0);
                    asm volatile("" : "+v"(v));
                    acc ^= v;          // (forces the wait: the register form is measured with the compiler's own vmcnt)
                }
                ++issued;
                if (MODE == 0) {
                    if (WINDOW == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else if (WINDOW == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    else if (WINDOW == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 0x12345678u && sink) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

template <int MODE, int WINDOW>
static void run(const char* name, const char* buf, long region_stride, int rows, int row_bytes, int cus, double ghz_hint, int iters = 1024) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(fetch_kernel<MODE, WINDOW>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((fetch_kernel<MODE, WINDOW>), dim3(cus), dim3(512), 131072, 0, buf, region_stride, rows, row_bytes, iters, (unsigned*)nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fetch_kernel<MODE, WINDOW>), dim3(cus), dim3(512), 131072, 0, buf, region_stride, rows, row_bytes, iters, (unsigned*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)cus * iters * (double)rows * row_bytes;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-34s window %2d: %7.3f ms  %6.2f TB/s chip-wide  %5.1f GB/s per CU  (~%4.1f B per clock and CU at %.1f GHz)\n", name, WINDOW, ms, tbs, tbs * 1e3 / cus, tbs * 1e12 / cus / (ghz_hint * 1e9), ghz_hint);
}

int main() {
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int rows = 256, row_bytes = 512;           // 128 KiB per region: 256 rows x 4 K tiles of 128 B
    const long priv = (long)rows * row_bytes;
    char* buf = nullptr;
    hipMalloc(&buf, (size_t)priv * cus + (1 << 20));
    hipMemset(buf, 1, (size_t)priv * cus + (1 << 20));
    const double ghz = 2.1;
    printf("fetch path probe: %d CUs, one 512-thread workgroup each, 1 KiB pieces (8 rows x 128 B), L2-resident regions\n", cus);
    run<0, 4>("dma, private lines", buf, priv, rows, row_bytes, cus, ghz);
    run<0, 8>("dma, private lines", buf, priv, rows, row_bytes, cus, ghz);
    run<0, 16>("dma, private lines", buf, priv, rows, row_bytes, cus, ghz);
    run<0, 32>("dma, private lines", buf, priv, rows, row_bytes, cus, ghz);
    run<0, 8>("dma, one region for all CUs", buf, 0, rows, row_bytes, cus, ghz);
    run<0, 16>("dma, one region for all CUs", buf, 0, rows, row_bytes, cus, ghz);
    run<1, 8>("registers, private lines", buf, priv, rows, row_bytes, cus, ghz);
    run<1, 8>("registers, one region for all CUs", buf, 0, rows, row_bytes, cus, ghz);
    hipFree(buf);
    // the same walk over lines that are NOT cache-resident: 4 MiB per CU (256 rows x 16 KiB), 1 GiB in all, each line touched once per pass
    const int big_row = 16384;
    const long big = (long)rows * big_row;
    hipMalloc(&buf, (size_t)big * cus + (1 << 20));
    hipMemset(buf, 1, (size_t)big * cus + (1 << 20));
    run<0, 4>("dma, streaming 1 GiB (HBM)", buf, big, rows, big_row, cus, ghz, 4);
    run<0, 8>("dma, streaming 1 GiB (HBM)", buf, big, rows, big_row, cus, ghz, 4);
    run<0, 16>("dma, streaming 1 GiB (HBM)", buf, big, rows, big_row, cus, ghz, 4);
    run<0, 32>("dma, streaming 1 GiB (HBM)", buf, big, rows, big_row, cus, ghz, 4);
    hipFree(buf);
    return 0;
}
