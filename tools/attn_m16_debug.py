"""Debug aid: flash_fwd3_kernel (SVI_FLASH_M16=1) against flash_fwd2_kernel (=0) and fp64 on one head; where does the difference sit?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
import svi_hip
from svi_hip import _lib as L
L_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn((1, L_, 128), generator=g, device="cuda").to(torch.bfloat16) for _ in range(3)]
def run(m16):
    L.set_switch("SVI_FLASH_M16", m16)
    o = svi_hip.flash_attention(q, k, v, 1)
    torch.cuda.synchronize()
    return o.float()[0]
o1 = run(1); o0 = run(0)
L.set_switch("SVI_FLASH_M16", None)
rows = torch.arange(0, L_, max(1, L_ // 512), device="cuda")
s = (q[0].double()[rows] @ k[0].double().t()) / 128 ** 0.5
ref = (torch.softmax(s, -1) @ v[0].double()).float()
def rel(a, b): return float((a - b).norm() / b.norm())
print("L", L_, "new vs fp64", rel(o1[rows], ref), "old vs fp64", rel(o0[rows], ref), "new vs old", rel(o1, o0))
d = (o1 - o0)
print("by 16-channel block:", [round(float(d[:, i * 16:(i + 1) * 16].norm() / o0[:, i * 16:(i + 1) * 16].norm()), 4) for i in range(8)])
print("by row % 64 in blocks of 16:", [round(float(d.view(-1, 4, 16, 128)[:, i].norm() / o0.view(-1, 4, 16, 128)[:, i].norm()), 4) for i in range(4)] if L_ % 64 == 0 else "-")
print("by channel % 16:", [round(float(d[:, i::16].norm() / o0[:, i::16].norm()), 4) for i in range(16)])
# which keys carry wrong weight: V = indicator of key % 128 (and of key // 128 for a second view)
eye = torch.zeros((1, L_, 128), device="cuda", dtype=torch.bfloat16)
eye[0, torch.arange(L_), torch.arange(L_) % 128] = 1
def runv(m16, vv):
    L.set_switch("SVI_FLASH_M16", m16)
    o = svi_hip.flash_attention(q, k, vv, 1); torch.cuda.synchronize(); return o.float()[0]
p1 = runv(1, eye); p0 = runv(0, eye)
L.set_switch("SVI_FLASH_M16", None)
dd = (p1 - p0)
print("indicator V (key % 128): rel", rel(p1, p0), "by key%64 groups of 8:", [round(float(dd[:, [c for c in range(128) if (c % 64) // 8 == i]].norm() / p0[:, [c for c in range(128) if (c % 64) // 8 == i]].norm()), 4) for i in range(8)])
pref = torch.softmax(s, -1).float()
pk = torch.zeros((rows.numel(), 128), device="cuda")
pk.index_add_(1, torch.arange(L_, device="cuda") % 128, pref)
print("indicator: new vs fp64", rel(p1[rows], pk), "old vs fp64", rel(p0[rows], pk))
