"""SURVEY §8f N1, the question behind a latent-space hand-off: the next clip's conditioning latent is VAE.encode([motion frame | 80 padding
frames]) (svi_video.py:329-350).  Could the encode be cut short — only the first K latent frames computed, the rest filled with the value the
causal encoder settles to on constant padding?  That would be EXACT only if the encoder's temporal receptive field (counted from the layer
table: 18 frames at full rate + 2 + 16 at half rate + 4 + 72 at quarter rate = 112 input frames) were shorter than the clip.  This tool measures
it on the HIP VAE at the C2 size: the difference between consecutive latent frames of encode([frame | zeros]); a settled encoder would show
exact zeros from some frame on.   python tools/n1_steady_state.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stable-video-infinity_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import svi_hip
from svi_hip.vae import WanVideoVAE, device_vae_weights
dev = torch.device("cuda")
vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
g = torch.Generator(device=dev).manual_seed(3)
for frames in (81, 161):
    vid = torch.zeros((3, frames, 480, 832), device=dev)
    vid[:, 0] = torch.tanh(torch.randn((3, 480, 832), generator=g, device=dev))
    z = vae.encode([vid], device=dev)[0]                     # [16, 1 + (frames-1)/4, 60, 104]
    d = (z[:, 1:] - z[:, :-1]).abs().amax(dim=(0, 2, 3))
    ref = z.abs().max().item()
    print(f"{frames} frames -> {z.shape[1]} latent frames; max |z[k+1] - z[k]| / max|z| per k:", " ".join(f"{v / ref:.1e}" for v in d.tolist()))
