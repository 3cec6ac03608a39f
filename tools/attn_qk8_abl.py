"""Timing-only ablations of the optimistic fp8 QK^T kernel (flash_fwd2_kernel<.., QK8>) at the C2 self-attention shape — a -DSVI_ABLATIONS variant build
(python tools/build_variant.py abl -DSVI_ABLATIONS; SVI_HIP_LIB=.../libsvi_hip_abl.so).  Results are WRONG for ABL != 0.
    python tools/attn_qk8_abl.py [rounds]
ABL: 0 the kernel; 1 no softmax work in the statements (exp / sum / pack); 8 no LDS-DMA (stale tiles); 264 = 8 + 256 no LDS-DMA and no workgroup barrier;
9 / 265 the combinations with 1.  'bf16' = the bf16 kernel in the same process."""
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
q_scaled = (q.float() * (1.4426950408889634 / 128 ** 0.5)).to(torch.bfloat16)   # what the DiT's RMSNorm+RoPE kernel hands over
L.set_switch("SVI_FLASH_ASSUME_PRESCALED", "1")
variants = (os.environ.get("ATTN_ABL_SET") or "bf16,0,1,8,264,9,265,0,bf16").split(",")
times = {a: [] for a in variants}
def run(a):
    L.set_switch("SVI_ATTN_QK8", "0" if a == "bf16" else "1")
    L.set_switch("SVI_FLASH_ABL", "0" if a == "bf16" else a)
    return svi_hip.flash_attention(q_scaled, k, v, n)
for a in variants:
    run(a)
torch.cuda.synchronize()
for _ in range(rounds):
    for a in dict.fromkeys(variants):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            run(a)
        e1.record(); torch.cuda.synchronize()
        times[a].append(e0.elapsed_time(e1) / 2)
fl = 4.0 * sq * sk * n * 128
for a in dict.fromkeys(variants):
    med = statistics.median(times[a])
    print(f"{a:>5}: med {med:.3f} ms  ({fl/med/1e9:.0f} TF-equivalent; seam incl. V transpose" + ("" if a == "bf16" else " + 2 quantiser launches") + f")  min {min(times[a]):.3f}")
