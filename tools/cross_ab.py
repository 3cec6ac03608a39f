"""Time the resident cross-attention kernel at the C2 size (L = 32760, 12 heads) for 33 / 65 / 97 / 128 keys, with and without the fused
q normalisation:   [SVI_HIP_LIB=<variant .so>] python tools/cross_ab.py [iters]
(no key_tail: the number of 32-key blocks then follows from Lk on the host, and the timed loop has no read-back in it).  Prints us per launch and the
fraction of 8 TB/s the q read + o write amount to."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch  # noqa: E402

import svi_hip  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
Ltok, D, H = 32760, 1536, 12


def rnd(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(torch.bfloat16)


q = rnd(Ltok, D, scale=2.0)
gain = rnd(D)
rs = (1.0 / torch.sqrt((q.float() ** 2).mean(-1) + 1e-6)).contiguous()
for Lk in (33, 65, 97, 128):
    k = rnd(Lk, D)
    ld = (Lk + 7) // 8 * 8
    vt = torch.zeros((D, ld), dtype=torch.bfloat16, device=dev)
    vt[:, :Lk] = rnd(D, Lk)
    for norm in (True, False):
        kw = dict(q_rs=rs, q_gain=gain, q_out_scale=0.12751743) if norm else {}
        fn = lambda: svi_hip.ops.cross_attention(q, k, vt, H, s_kv=Lk, **kw)  # noqa: E731
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"cross_ab lib={os.path.basename(os.environ.get('SVI_HIP_LIB') or 'libsvi_hip.so')} keys={Lk} norm={int(norm)} {us:.1f} us/launch "
              f"{4.0 * Ltok * D / us / 1e6 / 8.0:.3f} of 8 TB/s  finite={bool(torch.isfinite(out.float()).all())}")
