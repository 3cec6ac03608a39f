"""VAE decode / encode at a C1-like and the C2 size with each convolution family, in one process (switches re-read between runs):
    x2h    fp16 two-term convolution where the producer bounds the input, bf16 three-term elsewhere   (default)
    x3     bf16 three-term convolution everywhere                                                      (SVI_VAE_X2H=0)
    exact  fp32 MFMA everywhere                                                                        (SVI_VAE_EXACT_FP32=1; small size only unless `exact` is asked for)
Reports wall time, the per-tag kernel time (svi_prof_*) and the distance of each result to the exact-fp32 one (or to x3 at C2).
    python tools/vae_ab.py [c2] [exact] [only-default | ab-up | ab-order]
ab-up: the default against SVI_VAE_UP_PHASES=0 (upsample convolutions as one nine-tap convolution reading through the upsample).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import svi_hip
from svi_hip import _lib
from svi_hip.vae import WanVideoVAE, device_vae_weights

dev = torch.device("cuda")
c2 = "c2" in sys.argv[1:]
vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
g = torch.Generator(device=dev).manual_seed(3)
z = torch.randn((16, 21, 60, 104) if c2 else (16, 5, 32, 32), generator=g, device=dev)
vid = torch.tanh(torch.randn((3, 81, 480, 832) if c2 else (3, 17, 256, 256), generator=g, device=dev))
MODES = [("x2h", {}), ("x3", {"SVI_VAE_X2H": "0"})]
if not c2 or "exact" in sys.argv[1:]:
    MODES.append(("exact", {"SVI_VAE_EXACT_FP32": "1"}))
if "only-default" in sys.argv[1:]:             # for a kernel trace of the product path alone
    MODES = MODES[:1]
if "ab-up" in sys.argv[1:]:
    MODES = [("x2h", {}), ("up9", {"SVI_VAE_UP_PHASES": "0"})]
if "ab-pair" in sys.argv[1:]:
    MODES = [("pair", {}), ("single", {"SVI_VAE_PAIR": "0"}), ("pair", {})]
if "ab-order" in sys.argv[1:]:
    MODES = [("x2h", {}), ("pxord", {"SVI_VAE_TILE_ORDER": "0"}), ("x2h", {})]
KEYS = ("SVI_VAE_X2H", "SVI_VAE_EXACT_FP32", "SVI_VAE_UP_PHASES", "SVI_VAE_TILE_ORDER", "SVI_VAE_PAIR")
res = {}
for name, env in MODES:
    for k in KEYS:
        _lib.set_switch(k, env.get(k))
    for what, fn in (("decode", lambda: vae.decode([z], device=dev)[0]), ("encode", lambda: vae.encode([vid], device=dev)[0])):
        out = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.prof_enable(True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        tags = {k: round(v["ms"], 1) for k, v in _lib.prof_summary().items() if v["ms"] > 0}
        _lib.prof_enable(False)
        res[(name, what)] = out.double().cpu()
        print(f"{name:5s} {what} {tuple(out.shape)}: {e0.elapsed_time(e1):8.1f} ms   per-tag {tags}   absmax {float(out.abs().max()):.4f}", flush=True)
        del out
for k in KEYS:
    _lib.set_switch(k, None)
if "ab-pair" in sys.argv[1:]:
    for what in ("decode", "encode"):
        print(f"{what}: pair vs single bit-identical: {bool(torch.equal(res[('pair', what)], res[('single', what)]))}")
base = "exact" if ("exact", "decode") in res else "x3" if ("x3", "decode") in res else MODES[-1][0]
for what in ("decode", "encode") if len(MODES) > 1 else ():
    a = res[(base, what)]
    for name, _ in MODES:
        if name == base:
            continue
        b = res[(name, what)]
        print(f"{what}: {name} vs {base}: rel-L2 {float((a - b).norm() / a.norm()):.3e}  max-abs {float((a - b).abs().max()):.3e}")
