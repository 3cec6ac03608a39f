"""bf16 attention vs the opt-in fp8 QK^T attention (SVI_ATTN_QK8=1) at the C2 self-attention shape, interleaved in one process:
    python tools/attn_qk8_ab.py [rounds]          (SVI_HIP_LIB=<variant .so> to time a tools/build_variant.py build)
Times the public seam (svi_attention_fwd: V transpose + Q / K quantisation + kernel), prints medians and the distance between the two results."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
import svi_hip
from svi_hip import _lib as L
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda"); g = torch.Generator(device=dev).manual_seed(0)
sq = sk = 32760; n = 12
q, k, v = [(torch.randn((1, sq, n * 128), generator=g, device=dev)).to(torch.bfloat16) for _ in range(3)]
def run(mode):
    L.set_switch("SVI_ATTN_QK8", mode)
    return svi_hip.flash_attention(q, k, v, n)
a, b = run("0"), run("1")
torch.cuda.synchronize()
print(f"lib {os.environ.get('SVI_HIP_LIB', 'default')}: fp8 QK^T vs bf16 rel-L2 {float((a.float() - b.float()).norm() / a.float().norm()):.3e}")
times = {"0": [], "1": []}
for _ in range(rounds):
    for mode in ("0", "1"):
        run(mode); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run(mode)
        e1.record(); torch.cuda.synchronize()
        times[mode].append(e0.elapsed_time(e1) / 3)
fl = 4.0 * sq * sk * n * 128
for mode, name in (("0", "bf16"), ("1", "fp8 QK^T (incl. the two quantiser launches)")):
    med = statistics.median(times[mode])
    print(f"{name:46s}: med {med:.3f} ms ({fl / med / 1e9:.0f} TF incl. the V transpose of the seam)  min {min(times[mode]):.3f}")
L.set_switch("SVI_ATTN_QK8")
