"""Race screen for the staggered GEMM kernels: random shapes and epilogues, every 256-row tile kind against the 128^2 register-staged kernel, bit for bit, with and
without a bandwidth hog on a second stream (which stretches the LDS-DMA latencies the counted waits are placed by).  A clean screen proves nothing by itself (the
hand-off is argued from the vmcnt / barrier counts in the kernel's header); a dirty one would.   python tools/gemm_race_screen.py [iterations]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
from svi_hip import _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
rng = random.Random(1)
lib = L.lib()
hog_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
EPIS = [L.EPI_BIAS, L.EPI_BIAS_GELU_TANH, L.EPI_BIAS_GATE_RES, 4, 5]
bad, runs = 0, 0
for it in range(iters):
    M = rng.choice([rng.randint(1, 700), rng.randint(700, 6000), 256 * rng.randint(1, 40), 32760])
    N = 8 * rng.randint(1, 400) if rng.random() < 0.6 else 256 * rng.randint(1, 12)
    K = 64 * rng.randint(1, 40)
    if M * N * K > 6e11:
        M = max(1, int(6e11 / (N * K)))
    epi = rng.choice(EPIS)
    x = (torch.randn((M, K), generator=g, device=dev)).to(torch.bfloat16); w = (torch.randn((N, K), generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16); gate = torch.randn(N, generator=g, device=dev); res = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16)
    outs = {}
    hog = it % 2 == 1
    for kind in (128, 259, 260, 192) + ((264, 265) if os.environ.get("SVI_GEMM_EXPERIMENTS") else ()):
        if kind == 192 and N < 192:
            continue
        L.set_switch("SVI_GEMM_KERNEL", kind)
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
        if hog:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    hog_b.copy_(hog_a, non_blocking=True)
        L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, epi,
                                  gate.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, res.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, N, L.current_stream()))
        torch.cuda.synchronize()
        outs[kind] = out
    for kind, o in outs.items():
        runs += 1
        if not torch.equal(o, outs[128]):
            bad += 1
            print(f"MISMATCH kind {kind} M={M} N={N} K={K} epi={epi} hog={hog}: {int((o != outs[128]).sum())} elements", flush=True)
L.set_switch("SVI_GEMM_KERNEL")
print(f"{iters} shapes, {runs} launches compared with the 128^2 kernel, {bad} mismatching")
