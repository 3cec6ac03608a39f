"""The C1-size step's projections (M = 2560 stacked rows) alone: us per launch and TFLOP/s of svi_gemm_bf16 under the current switches / SVI_HIP_LIB.
    python tools/gemm_c1_probe.py [label]"""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stable-video-infinity_amd"))
import torch
import svi_hip as hip
L = hip._lib
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("SVI_HIP_LIB", "default")
M0 = int(os.environ.get("PROBE_M", "2560"))          # PROBE_M=65520: the C2-size stacked pair (tools/patches/gemm256_loop_ablation.patch builds are timed with it)
shapes = [("ffn2", 2560, 1536, 8960, L.EPI_BIAS_GATE_RES), ("attn_out", 2560, 1536, 1536, L.EPI_BIAS_GATE_RES), ("q", 2560, 1536, 1536, L.EPI_BIAS),
          ("ffn1", 2560, 8960, 1536, L.EPI_BIAS_GELU_TANH), ("qk480", 2560, 3072, 1536, L.EPI_BIAS), ("t720", 2560, 4608, 1536, L.EPI_BIAS)]
if os.environ.get("SVI_GEMM_KERNEL"):
    shapes = [s for s in shapes if s[0] != "ffn1"] + [("ffn1_128", 2560, 8960, 1536, L.EPI_BIAS_GELU_TANH)]
row = []
for name, M, N, K, epi in shapes:
    M = M0
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().cuda(); b = torch.randn(N, generator=g).bfloat16().cuda()
    kw = dict(epilogue=epi)
    if epi == L.EPI_BIAS_GATE_RES:
        kw.update(gate=torch.randn(N, generator=g).float().cuda(), residual=torch.randn(M, N, generator=g).bfloat16().cuda())
    for _ in range(5):
        hip.linear(x, w, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50 if M0 <= 4096 else 10
    e0.record()
    for _ in range(n):
        hip.linear(x, w, b, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    row.append(f"{name} {us:.1f} us ({2 * M * N * K / us / 1e6:.0f} TF)")
print(label, "|", " | ".join(row))
