import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
import svi_hip
from svi_hip import _lib as L
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
L.set_switch("SVI_FLASH_KERNEL", 2)
L_ = 64
g = torch.Generator(device="cuda").manual_seed(1)
q, k = [torch.randn((1, L_, 128), generator=g, device="cuda").to(torch.bfloat16) for _ in range(2)]
v = torch.zeros((1, L_, 128), device="cuda", dtype=torch.bfloat16)
v[0, torch.arange(L_), torch.arange(L_)] = 1
s = (q[0].double() @ k[0].double().t()) / 128 ** 0.5
pr = torch.softmax(s, -1).float()
L.set_switch("SVI_FLASH_M16", 1)
o = svi_hip.flash_attention(q, k, v, 1).float()[0][:, :64]; torch.cuda.synchronize()
ratio = o / pr
print("P(new)/P(ref), rows 44..63 (columns = keys 0..63, shown every key):")
for r in list(range(44, 64)):
    print(r, " ".join(f"{float(x):5.2f}" for x in ratio[r]))
L.set_switch("SVI_FLASH_M16", None); L.set_switch("SVI_FLASH_KERNEL", None)
