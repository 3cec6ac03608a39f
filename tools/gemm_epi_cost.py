"""What each fused epilogue costs on the ffn1 shape (M = 32760, N = 8960, K = 1536): the same GEMM with bias only / ReLU / SiLU / GELU-tanh / GELU-erf,
interleaved in one process.   python tools/gemm_epi_cost.py"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
from svi_hip import _lib as L
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 32760, 8960, 1536
lib = L.lib(); st = L.current_stream()
x = (torch.randn((M, K), generator=g, device=dev)).to(torch.bfloat16); w = (torch.randn((N, K), generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
EPIS = {"bias": L.EPI_BIAS, "relu": 5, "silu": 4, "gelu_tanh": L.EPI_BIAS_GELU_TANH, "gelu_erf": 3}
times = {k: [] for k in EPIS}
for _ in range(7):
    for name, epi in EPIS.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, epi, None, None, N, st))
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
print("  ".join(f"{k}: {statistics.median(v) * 1e3:.0f} us" for k, v in times.items()))
