"""VAE decode/encode at a C1-like and the C2 size: bf16x3 split convolution vs the exact-fp32 MFMA kernel (SVI_VAE_EXACT_FP32=1
in a separate process, the switch is read once).  python tools/vae_ab.py [c2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import svi_hip
from svi_hip.vae import WanVideoVAE, device_vae_weights
dev = torch.device("cuda")
vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
g = torch.Generator(device=dev).manual_seed(3)
shape = (16, 21, 60, 104) if len(sys.argv) > 1 and sys.argv[1] == "c2" else (16, 5, 32, 32)
z = torch.randn(shape, generator=g, device=dev)
for name, fn in (("decode", lambda: vae.decode([z], device=dev)),):
    out = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from svi_hip import _lib
    _lib.prof_enable(True)
    e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
    print("   per-tag:", {k: round(v["ms"], 1) for k, v in _lib.prof_summary().items()})
    _lib.prof_enable(False)
    print(f"{'exact-fp32' if os.environ.get('SVI_VAE_EXACT_FP32') else 'bf16x3'} {name} {tuple(shape)}: {e0.elapsed_time(e1):.1f} ms, checksum {float(out.double().sum()):.6f} absmax {float(out.abs().max()):.4f}")
    torch.save(out.cpu(), f"/tmp/vae_{name}_{'exact' if os.environ.get('SVI_VAE_EXACT_FP32') else 'x3'}.pt")
if os.path.exists("/tmp/vae_decode_exact.pt") and os.path.exists("/tmp/vae_decode_x3.pt"):
    a, b = torch.load("/tmp/vae_decode_exact.pt").double(), torch.load("/tmp/vae_decode_x3.pt").double()
    print(f"x3 vs exact-fp32: rel-L2 {float((a - b).norm() / a.norm()):.3e}  max-abs {float((a - b).abs().max()):.3e}")
