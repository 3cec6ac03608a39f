"""Timing-only ablations of flash_fwd2_kernel at the C2 self-attention shape (results are wrong for ABL != 0).
    python tools/attn_abl.py [rounds]
ABL bits: 1 no B fillers (exp/sum/pack), 2 no A fillers (row max), 4 fragment reads only at phase start, 8 no staging/barrier, 256 no workgroup barrier (vmcnt wait kept), 768 neither barrier nor vmcnt wait,
16 no finish/decision (max3 kept), 32 no max3 (finish kept), 64 no s_nop 15 at phase 2 start."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
import svi_hip
from svi_hip import _lib as L
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
sq = sk = 32760; n = 12
q, k, v = [(torch.randn((1, sq, n * 128), generator=g, device=dev)).to(torch.bfloat16) for _ in range(3)]
L.set_switch("SVI_FLASH_KERNEL", "2")
L.set_switch("SVI_FLASH_ASSUME_PRESCALED", "1")
variants = (os.environ.get("ATTN_ABL_SET") or "0,4,8,15,0").split(",")
times = {a: [] for a in variants}
variants_run = list(variants)
q_scaled = (q.float() * (1.4426950408889634 / 128 ** 0.5)).to(torch.bfloat16)   # what the DiT's RMSNorm+RoPE kernel hands over
def run(a):
    if a == "v1":
        L.set_switch("SVI_FLASH_KERNEL", "1")
    elif a == "mulc":
        L.set_switch("SVI_FLASH_ASSUME_PRESCALED"); L.set_switch("SVI_FLASH_KERNEL", "2"); L.set_switch("SVI_FLASH_ABL", "0")
    else:
        L.set_switch("SVI_FLASH_ASSUME_PRESCALED", "1")
        L.set_switch("SVI_FLASH_KERNEL", "2"); L.set_switch("SVI_FLASH_ABL", a)
    return svi_hip.flash_attention(q if a in ("v1", "mulc") else q_scaled, k, v, n)
for a in variants:
    run(a)
torch.cuda.synchronize()
for _ in range(rounds):
    for a in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            run(a)
        e1.record(); torch.cuda.synchronize()
        times[a].append(e0.elapsed_time(e1) / 2)
fl = 4.0 * sq * sk * n * 128
for a in variants:
    med = statistics.median(times[a])
    print(f"ABL={a:>3}: med {med:.3f} ms  ({fl/med/1e9:.0f} TF-equivalent)  min {min(times[a]):.3f}")
