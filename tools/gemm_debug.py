"""Where does a GEMM kernel variant differ from the reference variant?  python tools/gemm_debug.py A B  (SVI_GEMM_KERNEL values)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
from svi_hip import _lib as L
ka, kb = sys.argv[1], sys.argv[2]
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
lib = L.lib(); st = L.current_stream()
for (M, N, K) in ((32760, 1536, 1536), (2048, 1536, 1536), (32760, 1536, 256), (4096, 4096, 512)):
    x = (torch.randn((M, K), generator=g, device=dev)).to(torch.bfloat16); w = (torch.randn((N, K), generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    outs = {}
    for kind in (ka, kb):
        L.set_switch("SVI_GEMM_KERNEL", kind)
        o = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, o.data_ptr(), N, M, N, K, b.data_ptr(), 0, L.EPI_BIAS, None, None, 0, st))
        torch.cuda.synchronize(); outs[kind] = o.float()
    d = (outs[ka] != outs[kb])
    print(f"M={M} N={N} K={K}: mismatching {int(d.sum())} of {d.numel()}")
    if d.any():
        idx = d.nonzero()
        tm, tn = idx[:, 0] // 256, idx[:, 1] // 256
        tiles = torch.unique(tm * 1000 + tn)
        print("  tiles affected:", len(tiles), "of", ((M + 255) // 256) * ((N + 255) // 256), "first:", tiles[:12].tolist())
        r, c = idx[:, 0] % 256, idx[:, 1] % 256
        print("  rows-in-tile hist (per 32):", torch.bincount(r // 32, minlength=8).tolist(), " cols-in-tile hist (per 32):", torch.bincount(c // 32, minlength=8).tolist())
        ref = x.float() @ w.float().t() + b.float()
        ea, eb = (outs[ka] - ref).abs().max().item(), (outs[kb] - ref).abs().max().item()
        print(f"  max err vs fp32: {ka}: {ea:.3g}  {kb}: {eb:.3g}")
        t0 = int(tiles[0]); i0 = idx[(tm * 1000 + tn) == t0]
        print("  first bad tile", t0, "bad count", len(i0), "rows", sorted(set((i0[:, 0] % 256).tolist()))[:20], "cols", sorted(set((i0[:, 1] % 256).tolist()))[:20])
