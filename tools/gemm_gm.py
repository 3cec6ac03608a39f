"""Sweep the tile-group height (row panels per group) of the 256^2 GEMM on the C2 shapes.  python tools/gemm_gm.py"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
from svi_hip import _lib as L
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
Ltok, D, F = 32760, 1536, 8960
lib = L.lib(); st = L.current_stream()
def rnd(*shape, scale=1.0): return (torch.randn(shape, generator=g, device=dev) * scale).to(torch.bfloat16)
for name, (M, N, K, epi) in {"qkv": (Ltok, D, D, L.EPI_BIAS), "ffn1": (Ltok, F, D, L.EPI_BIAS_GELU_TANH), "ffn2": (Ltok, D, F, L.EPI_BIAS)}.items():
    x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = rnd(N); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    res = {}
    gms = ["1", "2", "4", "5", "6", "8", "16", "32"]
    times = {k: [] for k in gms}
    for _ in range(5):
        for gm in gms:
            L.set_switch("SVI_GEMM_GM", gm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, epi, None, None, N, st))
            e1.record(); torch.cuda.synchronize()
            times[gm].append(e0.elapsed_time(e1) / 3)
    print(name, " ".join(f"GM={gm}:{2.0*M*N*K/statistics.median(times[gm])/1e9:.0f}TF" for gm in gms), flush=True)
