"""Epilogue timing ablations of the 256^2 GEMM on the C2 shapes (SVI_GEMM_EPI_ABL: 0 full, 1 no epilogue, 2 no global stores,
3 stop after the LDS staging writes; results are wrong for != 0).   python tools/gemm_epi_abl.py"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
from svi_hip import _lib as L
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
Ltok, D, F = 32760, 1536, 8960
lib = L.lib(); st = L.current_stream()
rnd = lambda *s, scale=1.0: (torch.randn(s, generator=g, device=dev) * scale).to(torch.bfloat16)
SHAPES = {"ffn1 gelu": (Ltok, F, D, L.EPI_BIAS_GELU_TANH), "ffn2 gate+res": (Ltok, D, F, L.EPI_BIAS_GATE_RES),
          "qkv": (Ltok, D, D, L.EPI_BIAS), "attn_o gate+res": (Ltok, D, D, L.EPI_BIAS_GATE_RES)}
SETS = os.environ.get("ABL_SETS", "0,1,2,3").split(",")
for name, (M, N, K, epi) in SHAPES.items():
    x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = rnd(N); gate = torch.randn(N, generator=g, device=dev); res = rnd(M, N)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    def run():
        L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, epi,
                                  gate.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, res.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, N, st))
    times = {s: [] for s in SETS}
    for _ in range(5):
        for sset in SETS:
            L.set_switch("SVI_GEMM_EPI_ABL", sset)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): run()
            e1.record(); torch.cuda.synchronize()
            times[sset].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    print(f"{name:18s}", " | ".join(f"{s}: {statistics.median(t)*1e3:.0f}us {fl/statistics.median(t)/1e9:.0f}TF" for s, t in times.items()), flush=True)
