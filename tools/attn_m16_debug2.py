import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
import svi_hip
from svi_hip import _lib as L
L.set_switch("SVI_FLASH_KERNEL", 2)
for L_ in (64, 128, 192, 256, 320, 384, 512, 1024, 100, 1000):
    g = torch.Generator(device="cuda").manual_seed(L_)
    q, k, v = [torch.randn((1, L_, 128), generator=g, device="cuda").to(torch.bfloat16) for _ in range(3)]
    s = (q[0].double() @ k[0].double().t()) / 128 ** 0.5
    pr = torch.softmax(s, -1)
    ref = (pr @ v[0].double()).float()
    out = {}
    for m16 in (1, 0):
        L.set_switch("SVI_FLASH_M16", m16)
        o = svi_hip.flash_attention(q, k, v, 1); torch.cuda.synchronize()
        out[m16] = o.float()[0]
    def rel(a, b): return float((a - b).norm() / b.norm())
    n = min(L_, 64)
    byqb = [round(rel(out[1][i * 16:(i + 1) * 16], ref[i * 16:(i + 1) * 16]), 4) for i in range(min(4, (L_ + 15) // 16))]
    print(L_, "new", round(rel(out[1], ref), 5), "old", round(rel(out[0], ref), 5), "new by qb (wave 0):", byqb)
L.set_switch("SVI_FLASH_M16", None); L.set_switch("SVI_FLASH_KERNEL", None)
