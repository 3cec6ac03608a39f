"""Build container only (needs /root/reference): time ONE DiTBlock forward of the reference itself (models/wan_video_dit.py:354-374,
fp32, CPU) beside the oracle's restatement of it (oracle/wan_dit_oracle.py) on the same seeded weights and inputs at the full C2
token count, and the reference's VAE decode beside the oracle's on two latent frames at the C2 spatial size.  The ratios tie
bench.py's `cpu_baseline` (kind "port": the oracle timed on the GPU box, where the reference is absent) to the reference.

    python tools/ref_vs_oracle_block.py  > profiles/r2_cpu_reference_vs_port.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gen_golden  # noqa: E402
import synth  # noqa: E402
from oracle import wan_dit_oracle as wdo  # noqa: E402
from oracle import wan_vae_oracle as wvo  # noqa: E402

dit_mod, vae_mod, _ = gen_golden.import_reference()
t = gen_golden.t
threads = torch.get_num_threads()
grid = (21, 30, 52)
f, h, w = grid
L = f * h * w
c = dict(synth.WAN_1_3B, num_layers=1)
sd = synth.dit_state_dict(0, **c)
blk = dit_mod.DiTBlock(False, c["dim"], 12, c["ffn_dim"], 1e-6).eval()
blk.load_state_dict({k[len("blocks.0."):]: t(a) for k, a in sd.items() if k.startswith("blocks.0.")}, strict=True)
x = t(synth.randn(1, 1, L, 1536)); ctx = t(synth.randn(2, 1, 512, 1536)); tm = t(0.1 * synth.randn(3, 1, 6, 1536))
fr = dit_mod.precompute_freqs_cis_3d(128)
freqs = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                   fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
out = {"threads": threads, "tokens": L}
with torch.no_grad():
    t0 = time.time(); ref = blk(x, ctx, tm, freqs); out["reference_block_s"] = time.time() - t0
    sdt = {k: t(a) for k, a in sd.items() if k.startswith("blocks.0.")}
    t0 = time.time(); got = wdo.dit_block(sdt, "blocks.0.", x, ctx, tm, wdo.rope_table_3d(128, grid), wdo.DiTConfig(num_layers=1)); out["oracle_block_s"] = time.time() - t0
    out["block_rel_l2"] = float((got - ref).norm() / ref.norm())
    v = vae_mod.WanVideoVAE()
    vsd = {k: t(a) for k, a in synth.vae_state_dict(500).items()}
    v.load_state_dict(vsd, strict=True)
    z = t(synth.randn(511, 16, 2, 60, 104))
    t0 = time.time(); rv = v.decode([z], device="cpu")[0]; out["reference_vae_decode_2_latent_frames_s"] = time.time() - t0
    t0 = time.time(); ov = wvo.vae_decode(vsd, z[None])[0]; out["oracle_vae_decode_2_latent_frames_s"] = time.time() - t0
    out["vae_rel_l2"] = float((ov - rv).norm() / rv.norm())
out["port_over_reference_block"] = out["oracle_block_s"] / out["reference_block_s"]
out["port_over_reference_vae"] = out["oracle_vae_decode_2_latent_frames_s"] / out["reference_vae_decode_2_latent_frames_s"]
print(json.dumps(out, indent=1))
