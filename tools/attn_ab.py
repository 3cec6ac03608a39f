"""A/B the two flash-attention kernels (v1: 4 waves x 32 rows, 2 WG/CU; v2: 4 waves x 64 rows, pinned two-phase pipeline)
through the public seam svi_attention_fwd, interleaved in one process.

    python tools/attn_ab.py [rounds]
Correctness first (fp64 SDPA on the same bf16 inputs, several shapes incl. ragged tails and a forced-rescale spike),
then timing at the C2 self-attention shape (L = 32760, 12 heads) and the cross-attention shape (Lk = 512).
The kernel is chosen per launch by SVI_FLASH_KERNEL ("1" / "2").
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch  # noqa: E402

import svi_hip  # noqa: E402
from svi_hip import _lib as L  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(torch.bfloat16)


def ref(q, k, v, n):
    b, sq, dim = q.shape
    d = dim // n
    qq = q.double().view(b, sq, n, d).transpose(1, 2)
    kk = k.double().view(b, -1, n, d).transpose(1, 2)
    vv = v.double().view(b, -1, n, d).transpose(1, 2)
    p = torch.softmax(qq @ kk.transpose(-1, -2) / d ** 0.5, dim=-1)
    return (p @ vv).transpose(1, 2).reshape(b, sq, dim)


def run(kind, q, k, v, n):
    L.set_switch("SVI_FLASH_KERNEL", kind)
    return svi_hip.flash_attention(q, k, v, n)


ok = True
for (sq, sk, n, spike) in [(64, 64, 1, 0), (128, 512, 2, 0), (100, 77, 1, 0), (1280, 1280, 2, 0), (333, 257, 3, 0), (72, 72, 12, 0),
                           (1, 1, 1, 0), (4096, 4096, 1, 0), (130, 1000, 2, 0), (300, 129, 1, 0), (257, 64, 1, 0), (512, 640, 2, 1),
                           (700, 2100, 1, 2)]:
    q, k, v = rnd(1, sq, n * 128), rnd(1, sk, n * 128), rnd(1, sk, n * 128)
    if spike:                      # make one key dominate from a late tile on: forces the running-max rescale path
        pos = sk - 70 if spike == 1 else sk // 2
        k[0, pos] = (q[0, min(5, sq - 1)].float() * 4).to(torch.bfloat16)
    want = ref(q, k, v, n)
    line = f"Lq={sq:5d} Lk={sk:5d} heads={n:2d} spike={spike}:"
    for kind in ("1", "2"):
        got = run(kind, q, k, v, n).double()
        rel = float((got - want).norm() / want.norm())
        mx = float((got - want).abs().max())
        bad = not (rel < 3e-3) or not torch.isfinite(got).all()
        ok &= not bad
        line += f"  v{kind} rel {rel:.2e} max {mx:.2e}{' FAIL' if bad else ''}"
    print(line, flush=True)

for (sq, sk, n, name) in [(32760, 32760, 12, "self  L=32760"), (32760, 512, 12, "cross Lk=512")]:
    q, k, v = rnd(1, sq, n * 128), rnd(1, sk, n * 128), rnd(1, sk, n * 128)
    o1 = run("1", q, k, v, n); o2 = run("2", q, k, v, n)
    torch.cuda.synchronize()
    d = (o1.float() - o2.float())
    print(f"{name}: v1 vs v2 rel {float(d.norm() / o1.float().norm()):.2e} max {float(d.abs().max()):.2e}", flush=True)
    times = {"1": [], "2": []}
    for _ in range(rounds):
        for kind in ("1", "2"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(kind, q, k, v, n)
            e1.record(); torch.cuda.synchronize()
            times[kind].append(e0.elapsed_time(e1) / 3)
    fl = 4.0 * sq * sk * n * 128
    msg = name
    for kind in ("1", "2"):
        med, mn = statistics.median(times[kind]), min(times[kind])
        msg += f" | v{kind}: med {med:.3f} ms {fl/med/1e9:.0f} TF, best {fl/mn/1e9:.0f} TF (incl. V transpose)"
    print(msg, flush=True)
L.set_switch("SVI_FLASH_KERNEL")
print("ALL OK" if ok else "FAILURES")
