"""Which bf16 MFMA shape delivers more under the part's POWER limit?   python tools/probe/run_mfma_power_probe.py [build]

Round-6 question: this repo's GEMM and attention kernels multiply with v_mfma_f32_32x32x16_bf16; the vendor library's
winning GEMM for the same shapes is built on v_mfma_f32_16x16x32_bf16 (kernel name ..._MT256x256x64_MI16x16x1_..., read from a rocprofv3
trace).  Both shapes have the same nominal rate.  On random operands every matrix kernel here is held by the package power limit, not by
its instruction stream, so the shape that costs fewer joules per flop wins.  This probe takes memory out of the picture: one wave per SIMD,
a register-resident 128 x 128 outer product per wave (what a 256-thread 256^2 tile gives a wave), operands fixed in VGPRs, accumulators in
AGPRs, nothing but MFMAs in the loop:
    s32:  4 A x 4 B fragments,  16 accumulators of 16 registers, two k-steps of 16  -> 32 x v_mfma_f32_32x32x16_bf16 per iteration
    s16:  8 A x 8 B fragments,  64 accumulators of  4 registers, one k-step of 32   -> 64 x v_mfma_f32_16x16x32_bf16 per iteration
(same flop per iteration, same operand registers).  Each is run on zeros and on unit-Gaussian operands for a few seconds while rocm-smi is
sampled; the sustained TFLOP/s of the second half of the run is reported.
"""
import ctypes as C, os, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mfma_power_probe.hip")
SO = os.path.join(HERE, "libmfma_power_probe.so")


def gen() -> str:
    agprs = ", ".join(f'"a{i}"' for i in range(256))
    zero = "\\n\\t".join(f"v_accvgpr_write_b32 a{i}, 0" for i in range(256))
    # s32: acc(i, j) = a[16 (4 i + j) : +15];  A fragment i of k-step kk = %[a{4 kk + i}], B likewise (8 fragments of each kind in all)
    s32 = []
    for kk in range(2):
        for i in range(4):
            for j in range(4):
                r = 16 * (4 * i + j)
                s32.append(f"v_mfma_f32_32x32x16_bf16 a[{r}:{r + 15}], %[a{4 * kk + i}], %[b{4 * kk + j}], a[{r}:{r + 15}]")
    s16 = []
    for i in range(8):
        for j in range(8):
            r = 4 * (8 * i + j)
            s16.append(f"v_mfma_f32_16x16x32_bf16 a[{r}:{r + 3}], %[a{i}], %[b{j}], a[{r}:{r + 3}]")
    # s16 with the vendor loop's order (B fastest) is what is above; s16t walks A fastest (the other operand toggles between neighbours)
    s16t = []
    for j in range(8):
        for i in range(8):
            r = 4 * (8 * i + j)
            s16t.append(f"v_mfma_f32_16x16x32_bf16 a[{r}:{r + 3}], %[a{i}], %[b{j}], a[{r}:{r + 3}]")
    # attention-like mixes (flash_fwd2_kernel's balanced optimistic schedule carries, per 32x32x16 MFMA, one score: v_exp_f32 + v_add_f32 + half a
    # v_cvt_pk_bf16_f32, and half a ds_read_b128): the same VALU / LDS work per flop beside either shape
    def mix(lines, per):          # per = MFMAs per (exp, add), 2 per = per cvt and per ds_read
        out = []
        for n, l in enumerate(lines):
            if n % per == 0:
                out.append("v_exp_f32 %[t0], %[x0]")
            out.append(l)
            if n % per == per - 1:
                out.append("v_add_f32 %[s0], %[s0], %[t0]")
            if n % (2 * per) == 2 * per - 1:
                out.append("v_cvt_pk_bf16_f32 %[w0], %[t0], %[s0]")
                out.append("ds_read_b128 %[fr], %[ad]")
        out.append("s_waitcnt lgkmcnt(0)")
        return out

    def mix_fine(lines):          # 16x16x32 with the same work spread one piece per MFMA gap: exp | M | add | M | exp | M | add, cvt or ds_read | M
        out = []
        for n, l in enumerate(lines):
            if n % 2 == 0:
                out.append("v_exp_f32 %[t0], %[x0]")
            else:
                out.append("v_add_f32 %[s0], %[s0], %[t0]")
                if n % 8 == 3:
                    out.append("v_cvt_pk_bf16_f32 %[w0], %[t0], %[s0]")
                if n % 8 == 7:
                    out.append("ds_read_b128 %[fr], %[ad]")
            out.append(l)
            if n % 8 == 5:
                out.append("v_cvt_pk_bf16_f32 %[w0], %[t0], %[x0]")
        out.append("s_waitcnt lgkmcnt(0)")
        return out
    def mix_pk(lines):            # as mix_fine, the pair's two row-sum adds as ONE v_pk_add_f32 on fixed register pairs: exp | M | - | M | exp | M | pk_add, cvt | M
        out = []
        for n, l in enumerate(lines):
            if n % 4 == 0:
                out.append("v_exp_f32 v200, %[x0]")
            if n % 4 == 2:
                out.append("v_exp_f32 v201, %[x0]")
            if n % 4 == 3:
                out.append("v_pk_add_f32 v[202:203], v[202:203], v[200:201]")
                out.append("v_cvt_pk_bf16_f32 %[w0], v200, v201")
            if n % 8 == 5:
                out.append("ds_read_b128 %[fr], %[ad]")
            out.append(l)
        out.append("s_waitcnt lgkmcnt(0)")
        return out
    ins = ", ".join([f'[a{i}] "v"(a[{i}])' for i in range(8)] + [f'[b{i}] "v"(b[{i}])' for i in range(8)])

    def body(lines):
        return '"' + "\\n\\t".join(lines) + '"'
    return f"""// generated by run_mfma_power_probe.py — do not edit
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* __restrict__ opa, const u32x4* __restrict__ opb, float* __restrict__ out, int loops) {{
    u32x4 a[8], b[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < 8; ++i) {{ a[i] = opa[(wave * 8 + i) * 64 + lane]; b[i] = opb[(wave * 8 + i) * 64 + lane]; }}
    asm volatile("{zero}" ::: {agprs});
    __shared__ u32x4 lds[1024];
    lds[threadIdx.x] = a[0]; lds[threadIdx.x + 256] = a[1]; lds[threadIdx.x + 512] = b[0]; lds[threadIdx.x + 768] = b[1];
    __syncthreads();
    float t0, s0 = 0.f, x0 = -1.0f - 0.01f * lane;
    unsigned w0;
    u32x4 fr;
    const int ad = (int)(threadIdx.x * 16);
    for (int it = 0; it < loops; ++it) {{
        if constexpr (SHAPE == 32) asm volatile({body(s32)} :: {ins} : {agprs});
        else if constexpr (SHAPE == 16) asm volatile({body(s16)} :: {ins} : {agprs});
        else if constexpr (SHAPE == 17) asm volatile({body(s16t)} :: {ins} : {agprs});
        else if constexpr (SHAPE == 132) asm volatile({body(mix(s32, 1))} : [t0] "=&v"(t0), [s0] "+v"(s0), [w0] "=&v"(w0), [fr] "=&v"(fr) : {ins}, [x0] "v"(x0), [ad] "v"(ad) : {agprs});
        else if constexpr (SHAPE == 118) asm volatile({body(mix_pk(s16))} : [t0] "=&v"(t0), [s0] "+v"(s0), [w0] "=&v"(w0), [fr] "=&v"(fr) : {ins}, [x0] "v"(x0), [ad] "v"(ad) : {agprs}, "v200", "v201", "v202", "v203");
        else if constexpr (SHAPE == 117) asm volatile({body(mix_fine(s16))} : [t0] "=&v"(t0), [s0] "+v"(s0), [w0] "=&v"(w0), [fr] "=&v"(fr) : {ins}, [x0] "v"(x0), [ad] "v"(ad) : {agprs});
        else asm volatile({body(mix(s16, 2))} : [t0] "=&v"(t0), [s0] "+v"(s0), [w0] "=&v"(w0), [fr] "=&v"(fr) : {ins}, [x0] "v"(x0), [ad] "v"(ad) : {agprs});
    }}
    float x;
    asm volatile("s_nop 7\\n\\ts_nop 7\\n\\tv_accvgpr_read_b32 %0, a5" : "=v"(x) :: {agprs});
    if (x == 123.456f || s0 == 123.456f) out[threadIdx.x] = x + s0 + (float)w0 + (float)fr[0];
}}
extern "C" int probe_run(int shape, int blocks, int loops, const void* opa, const void* opb, void* out, float* ms) {{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    else if (shape == 16) hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    else if (shape == 132) hipLaunchKernelGGL(probe<132>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    else if (shape == 118) hipLaunchKernelGGL(probe<118>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    else if (shape == 117) hipLaunchKernelGGL(probe<117>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    else if (shape == 116) hipLaunchKernelGGL(probe<116>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    else hipLaunchKernelGGL(probe<17>, dim3(blocks), dim3(256), 0, 0, (const u32x4*)opa, (const u32x4*)opb, (float*)out, loops);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 1;
    hipEventElapsedTime(ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (int)hipGetLastError();
}}
"""


if len(sys.argv) > 1 and sys.argv[1] == "build":
    open(SRC, "w").write(gen())
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", SRC, "-o", SO], check=True)
    print("built", SO)
    sys.exit(0)

import torch
lib = C.CDLL(SO)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
FLOP_PER_BLOCK_ITER = 4 * 64 * 2.0 * 16 * 16 * 32          # four waves x 64 MFMAs of 16x16x32 (= 32 of 32x32x16)
BLOCKS = 256 * 8
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power" in l and "W" in l.split(":")[-1] or "Power (W)" in l]
            sc = [l.split("(")[-1].split(")")[0] for l in out.splitlines() if "sclk" in l]
            samples.append((time.time(), pw[:1], sc[:1]))
        except Exception as ex:
            samples.append((time.time(), [repr(ex)], []))
        time.sleep(0.25)


th = threading.Thread(target=sampler)
th.start()
out = torch.zeros(1024, device=dev)
ms = C.c_float()
rows = []
try:
    for data in ("zeros", "gauss"):
        sc = {"zeros": 0.0, "gauss": 1.0, "gauss_1e-3": 1e-3}[data]
        opa = (torch.randn((4 * 8 * 64, 8), generator=g, device=dev) * sc).to(torch.bfloat16).contiguous()
        opb = (torch.randn((4 * 8 * 64, 8), generator=g, device=dev) * sc).to(torch.bfloat16).contiguous()
        for shape, name in ((32, "32x32x16"), (16, "16x16x32 (B fastest)"), (17, "16x16x32 (A fastest)"), (32, "32x32x16 (again)"),
                            (132, "32x32x16 + softmax mix"), (116, "16x16x32 + softmax mix"), (117, "16x16x32 + mix, one piece per gap"), (118, "16x16x32 + mix, row sums by v_pk_add_f32"), (132, "32x32x16 + mix (again)")):
            loops = 2000
            lib.probe_run(shape, BLOCKS, loops, C.c_void_p(opa.data_ptr()), C.c_void_p(opb.data_ptr()), C.c_void_p(out.data_ptr()), C.byref(ms))
            loops = max(500, int(loops * 350.0 / max(ms.value, 1e-3)))          # ~0.35 s per launch
            t_begin = time.time()
            tf = []
            while time.time() - t_begin < 3.0:
                rc = lib.probe_run(shape, BLOCKS, loops, C.c_void_p(opa.data_ptr()), C.c_void_p(opb.data_ptr()), C.c_void_p(out.data_ptr()), C.byref(ms))
                if rc:
                    raise RuntimeError(f"probe_run {rc}")
                tf.append((time.time(), BLOCKS * loops * FLOP_PER_BLOCK_ITER / (ms.value * 1e-3) / 1e12))
            t_end = time.time()
            late = [v for t, v in tf if t - t_begin > 1.5] or [tf[-1][1]]
            smp = [s for s in samples if t_begin + 1.5 < s[0] < t_end]
            rows.append((data, name, sum(late) / len(late), tf[0][1], smp[-1][1:] if smp else None))
            print(f"{data:11s} {name:24s} sustained {rows[-1][2]:7.1f} TFLOP/s   first launch {tf[0][1]:7.1f}   rocm-smi {rows[-1][4]}", flush=True)
finally:
    stop = True
    th.join()
