"""Generate tests/golden/*.npz by running the REFERENCE's own modules on seeded inputs.

Runs only where /root/reference exists (the build container).  The reference is *imported*
(never copied) through the namespace-stub recipe of SURVEY.md Appendix A; outputs are stored as
small fixtures so that the oracle and the HIP path can be checked anywhere, including on the GPU
box where the reference is absent.

    python tests/gen_golden.py            # rewrites every fixture

Weights and inputs are NOT stored: tests/synth.py regenerates them bit-identically from seeds.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

REF = os.environ.get("SVI_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "golden")


def import_reference():
    """The reference's modules, imported where they lie under /root/reference (oracle/ref_shim.py: SURVEY.md Appendix A)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import ref_shim
    return ref_shim.load(REF)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_ref_dit(dit_mod, cfg, seed):
    m = dit_mod.WanModel(eps=1e-6, num_heads=synth.num_heads_of(cfg), **cfg).eval()
    sd = {k: t(v) for k, v in synth.dit_state_dict(seed, **cfg).items()}
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m


def gen_flow_match(fm):
    out = {}
    for n in (10, 50, 4):
        s = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(n, shift=5.0)
        out[f"sigmas_{n}"] = s.sigmas.numpy()
        out[f"timesteps_{n}"] = s.timesteps.numpy()
        # Euler update of a known sample through every step
        x = t(synth.randn(7, 2, 3))
        v = t(synth.randn(8, 2, 3))
        traj = []
        for i in range(n):
            x = s.step(v, s.timesteps[i], x)
            traj.append(x.numpy().copy())
        out[f"euler_traj_{n}"] = np.stack(traj)
    np.savez(os.path.join(OUT, "flow_match.npz"), **out)


def dit_case(dit_mod, name, cfg, grid, ctx_tokens, ctx_valid, timestep, seed):
    f, h, w = grid
    m = build_ref_dit(dit_mod, cfg, seed)
    x = t(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    ctx = t(synth.text_context(seed + 2, ctx_tokens, cfg["text_dim"], ctx_valid))
    ts = torch.tensor([timestep], dtype=torch.float32)
    kw = {}
    if cfg["has_image_input"]:
        kw["clip_feature"] = t(synth.randn(seed + 3, 1, 257, 1280))
        kw["y"] = t(synth.randn(seed + 4, 1, cfg["in_dim"] - 16, f, 2 * h, 2 * w))
    out = {}
    with torch.no_grad():
        out["out_fp32"] = m(x, ts, ctx, **kw).numpy()
        # one block in isolation (DiTBlock.forward), on its own seeded inputs
        L = f * h * w
        bx = t(synth.randn(seed + 5, 1, L, cfg["dim"]))
        bctx = t(synth.randn(seed + 6, 1, ctx_tokens + (257 if cfg["has_image_input"] else 0), cfg["dim"]))
        btm = t(0.5 * synth.randn(seed + 7, 1, 6, cfg["dim"]))
        freqs = torch.cat([
            m.freqs[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
            m.freqs[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            m.freqs[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
        out["block0_fp32"] = m.blocks[0](bx, bctx, btm, freqs).numpy()
        out["rope_table"] = torch.view_as_real(freqs[:, 0]).numpy()          # [L, dh/2, 2] fp64
        # the same model the way the pipelines run it: bf16 weights and activations
        mb = build_ref_dit(dit_mod, cfg, seed).to(torch.bfloat16)
        kwb = {k: v.to(torch.bfloat16) for k, v in kw.items()}
        out["out_bf16"] = mb(x.to(torch.bfloat16), ts, ctx.to(torch.bfloat16), **kwb).float().numpy()
        out["block0_bf16"] = mb.blocks[0](bx.to(torch.bfloat16), bctx.to(torch.bfloat16),
                                          btm.to(torch.bfloat16), freqs).float().numpy()
    np.savez(os.path.join(OUT, f"dit_{name}.npz"), **out)
    return m


def gen_denoise(dit_mod, fm):
    """4-step CFG denoise of a tiny clip with the reference scheduler + model (svi_video.py:392-421 loop)."""
    cfg, seed, grid = synth.TINY_DIT, 300, (2, 4, 4)
    m = build_ref_dit(dit_mod, cfg, seed)
    f, h, w = grid
    g = torch.Generator("cpu").manual_seed(11)
    lat = torch.randn((1, 16, f, 2 * h, 2 * w), generator=g, dtype=torch.float32)
    pos = t(synth.text_context(seed + 2, 16, cfg["text_dim"], 9))
    neg = t(synth.text_context(seed + 3, 16, cfg["text_dim"], 4))
    s = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    s.set_timesteps(4, shift=5.0)
    with torch.no_grad():
        for i, ts in enumerate(s.timesteps):
            tt = ts.unsqueeze(0)
            c = m(lat, tt, pos)
            u = m(lat, tt, neg)
            lat = s.step(u + 5.0 * (c - u), s.timesteps[i], lat)
    np.savez(os.path.join(OUT, "denoise_tiny.npz"), latents=lat.numpy(), noise_head=torch.randn(
        (8,), generator=torch.Generator("cpu").manual_seed(11)).numpy())


def gen_vae(vae_mod):
    v = vae_mod.WanVideoVAE()
    sd = {k: t(a) for k, a in synth.vae_state_dict(500).items()}
    v.load_state_dict(sd, strict=True)
    out = {}
    with torch.no_grad():
        z = t(synth.randn(501, 1, 16, 3, 4, 6))
        out["decode_3f"] = v.decode([z[0]], device="cpu")[0].numpy()               # [3,9,32,48]
        z1 = t(synth.randn(502, 1, 16, 1, 4, 6))
        out["decode_1f"] = v.decode([z1[0]], device="cpu")[0].numpy()              # [3,1,32,48]
        vid = t(np.tanh(synth.randn(503, 3, 9, 32, 48)))
        out["encode_9f"] = v.encode([vid], device="cpu")[0].numpy()                # [16,3,4,6]
        vid1 = t(np.tanh(synth.randn(504, 3, 1, 32, 48)))
        out["encode_1f"] = v.encode([vid1], device="cpu")[0].numpy()               # [16,1,4,6]
        z5 = t(synth.randn(505, 1, 16, 2, 2, 2))
        out["decode_2f_tinyhw"] = v.decode([z5[0]], device="cpu")[0].numpy()       # [3,5,16,16]
    np.savez(os.path.join(OUT, "vae.npz"), **out)


def _reference_method(rel_path, class_name, func_name, namespace):
    """Compile ONE method of a reference class out of its source file (the pipeline modules cannot be imported whole here:
    transformers/ftfy/imageio...) and return it as a plain function.  The reference code is executed, not copied."""
    import ast
    tree = ast.parse(open(os.path.join(REF, rel_path)).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == func_name:
                    mod = ast.Module(body=[fn], type_ignores=[])
                    exec(compile(mod, os.path.join(REF, rel_path), "exec"), namespace)
                    return namespace[func_name]
    raise KeyError((rel_path, class_name, func_name))


def _reference_toplevel(rel_path, name, namespace):
    """Compile ONE top-level class or function of a reference source file (see _reference_method)."""
    import ast
    tree = ast.parse(open(os.path.join(REF, rel_path)).read())
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name == name:
            exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF, rel_path), "exec"), namespace)
            return namespace[name]
    raise KeyError((rel_path, name))


TEA_STEPS, TEA_THRESH, TEA_MODEL, TEA_TIMESTEPS = synth.TEA_STEPS, synth.TEA_THRESH, synth.TEA_MODEL, synth.TEA_TIMESTEPS


def gen_teacache(dit_mod, fm):
    """The reference's own TeaCache class and model_fn_wan_video (compiled out of pipelines/svi_video.py) driving the reference
    WanModel (tiny T2V config, seeded) through an 8-step single-branch loop: which steps skip, and every step's output."""
    ns = {"torch": torch, "np": np, "WanModel": dit_mod.WanModel, "Optional": None,
          "sinusoidal_embedding_1d": dit_mod.sinusoidal_embedding_1d}
    TeaCache = _reference_toplevel("diffsynth/pipelines/svi_video.py", "TeaCache", ns)
    ns["TeaCache"] = TeaCache
    model_fn = _reference_toplevel("diffsynth/pipelines/svi_video.py", "model_fn_wan_video", ns)
    c, grid, nt, nv, seed = synth.TINY_DIT, (3, 4, 6), 20, 13, 100
    m = dit_mod.WanModel(eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    m.load_state_dict({k: t(a) for k, a in synth.dit_state_dict(seed, **c).items()}, strict=True)
    m = m.to(torch.bfloat16).eval()
    f, h, w = grid
    x = t(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w)).to(torch.bfloat16)
    ctx = t(synth.text_context(seed + 2, nt, c["text_dim"], nv)).to(torch.bfloat16)
    # Random-init time embeddings are chaotic in t (cos(t) itself turns by radians per unit step), so neighbouring rungs of a
    # real ladder would always look "far apart"; closely spaced timesteps give the small, varied t_mod changes that make the
    # accumulate / threshold / reset logic take both branches.
    timesteps = torch.tensor(TEA_TIMESTEPS, dtype=torch.float32)
    tc = TeaCache(TEA_STEPS, rel_l1_thresh=TEA_THRESH, model_id=TEA_MODEL)
    outs, skipped, tmods = [], [], []
    with torch.no_grad():
        for i in range(TEA_STEPS):
            ts = timesteps[i:i + 1]
            before = tc.previous_residual
            o = model_fn(m, x, timestep=ts, context=ctx, tea_cache=tc)
            skipped.append(tc.previous_hidden_states is None and before is tc.previous_residual and i not in (0, TEA_STEPS - 1))
            outs.append(o.float().numpy())
            tmods.append(tc.previous_modulated_input.float().numpy())
            x = (x.float() + 0.05 * o.float()).to(torch.bfloat16)          # any deterministic latent update keeps the loop moving
    np.savez(os.path.join(OUT, "teacache_tiny.npz"), outs=np.stack(outs), skipped=np.array(skipped), t_mod=np.stack(tmods),
             timesteps=timesteps.numpy())
    print("teacache skipped:", skipped)


def gen_image_condition(vae_mod):
    """SVIVideoPipeline.encode_images_adaptive (pipelines/svi_video.py:291-364) run on a stand-in `self` that carries the
    reference VAE (seeded weights), BasePipeline.preprocess_image and the reference's image encoder at a tiny configuration: pins row a22
    (y = mask | VAE latent) and the clip_feature the same call returns."""
    from PIL import Image
    ns = {"torch": torch, "np": np}
    encode_images_adaptive = _reference_method("diffsynth/pipelines/svi_video.py", "SVIVideoPipeline", "encode_images_adaptive", ns)
    preprocess_image = _reference_method("diffsynth/pipelines/base.py", "BasePipeline", "preprocess_image", ns)
    v = vae_mod.WanVideoVAE()
    v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)

    # the image encoder: the reference's own VisionTransformer (tiny configuration, seeded weights) behind the reference's own
    # WanImageEncoder.encode_image (compiled out of the source file; torchvision's Normalize stated as in gen_clip)
    import torch.nn.functional as F
    ie = importlib.import_module("diffsynth.models.wan_video_image_encoder")
    ccfg = synth.CLIP_TINY
    vis = ie.VisionTransformer(image_size=ccfg["image_size"], patch_size=ccfg["patch_size"], dim=ccfg["dim"], mlp_ratio=ccfg["mlp_ratio"], out_dim=64,
                               num_heads=ccfg["num_heads"], num_layers=ccfg["num_layers"], pool_type="token", pre_norm=True, post_norm=False,
                               activation="gelu", norm_eps=1e-5).eval()
    vis.load_state_dict({k: t(a) for k, a in synth.clip_state_dict(synth.CLIP_SEED, **ccfg).items()}, strict=True)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)

    class Clip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = torch.nn.Module()
            self.model.visual = vis
            self.model.image_size = ccfg["image_size"]
            self.transforms = types.SimpleNamespace(transforms=[lambda x: x.sub_(mean).div_(std)])

    Clip.encode_image = _reference_method("diffsynth/models/wan_video_image_encoder.py", "WanImageEncoder", "encode_image", {"torch": torch, "F": F})

    class Self:
        torch_dtype = torch.bfloat16
        device = "cpu"

    me = Self()
    me.vae, me.image_encoder = v, Clip()
    me.preprocess_image = lambda image, use_aug=False: preprocess_image(me, image, use_aug)
    H, W, T = 32, 48, 9
    out = {}
    with torch.no_grad():
        for i, (name, n, cfg, pad) in enumerate(synth.IMAGE_CONDITION_CASES):
            frames = [Image.fromarray(a) for a in synth.condition_frames(600 + i, n, H, W)]
            ref = Image.fromarray(synth.condition_frames(650 + i, 1, H, W)[0])
            r = encode_images_adaptive(me, frames, ref, T, H, W, use_first_aug=False, ref_pad_cfg=cfg, ref_pad_num=pad)
            assert r["y"].dtype == torch.bfloat16 and tuple(r["y"].shape) == (1, 20, 3, 4, 6)
            assert r["clip_feature"].dtype == torch.bfloat16 and tuple(r["clip_feature"].shape) == (1, 5, ccfg["dim"])
            out[name] = r["y"].float().numpy()
            out["clip_" + name] = r["clip_feature"].float().numpy()
    np.savez(os.path.join(OUT, "image_condition.npz"), **out)


# ------------------------------------------------------------------------------------------------------------------
# Fixtures at the BASELINE configurations' own sizes.  These run the reference for minutes, so
# `python tests/gen_golden.py <name> ...` regenerates only the named fixtures (no argument: all of them).
# ------------------------------------------------------------------------------------------------------------------
C1_VIDEO_STRIDE = synth.C1_VIDEO_STRIDE


def gen_c1_e2e(dit_mod, vae_mod, fm):
    """BASELINE config 1 end to end on the reference itself: the full 30-layer Wan2.1-T2V-1.3B architecture (seeded weights),
    latent [1,16,5,32,32] (17 frames 256x256), 10 flow-match steps with CFG 5 — the loop of svi_video.py:392-421 on
    WanModel.forward — then WanVideoVAE.decode in fp32 (svi_video.py:384-389).  fp32 run = the value reference; the same loop the
    way the pipelines run it (bf16 weights, activations and latents) gives the reference's own bf16-vs-fp32 gap as the yardstick."""
    import time
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    t0 = time.time()
    m = build_ref_dit(dit_mod, cfg, seed)
    print(f"c1: reference WanModel 1.3B built in {time.time() - t0:.0f} s")
    noise = torch.randn((1, 16, 5, 32, 32), generator=torch.Generator("cpu").manual_seed(0), dtype=torch.float32)   # base.py:140-143
    pos = t(synth.text_context(seed + 1, 512, cfg["text_dim"], 64))
    neg = t(synth.text_context(seed + 2, 512, cfg["text_dim"], 64))

    def loop(model, lat, pos, neg):
        s = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(10, shift=5.0)
        with torch.no_grad():
            for i, ts in enumerate(s.timesteps):
                tt = ts.unsqueeze(0)
                c = model(lat, tt, pos)
                u = model(lat, tt, neg)
                lat = s.step(u + 5.0 * (c - u), s.timesteps[i], lat)
        return lat

    t0 = time.time()
    lat32 = loop(m, noise, pos, neg)
    print(f"c1: fp32 loop {time.time() - t0:.0f} s")
    t0 = time.time()
    mb = m.to(torch.bfloat16)
    lat16 = loop(mb, noise.to(torch.bfloat16), pos.to(torch.bfloat16), neg.to(torch.bfloat16)).float()
    print(f"c1: bf16 loop {time.time() - t0:.0f} s")
    del m, mb
    v = vae_mod.WanVideoVAE()
    v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)
    with torch.no_grad():
        t0 = time.time()
        video = v.decode([lat32[0]], device="cpu")[0]                                  # [3,17,256,256] fp32, clamped
        print(f"c1: VAE decode {time.time() - t0:.0f} s")
    k = C1_VIDEO_STRIDE
    np.savez(os.path.join(OUT, "c1_e2e.npz"), latents_fp32=lat32[0].numpy(), latents_bf16=lat16[0].numpy(),
             video_sample=video[:, :, ::k, ::k].contiguous().numpy(), video_shape=np.array(video.shape),
             video_absmean=np.array(float(video.abs().mean())))


def gen_c1_50step(dit_mod, fm):
    """The headline's HORIZON on the reference itself: the C1 latent grid [1,16,5,32,32] (1 280 tokens), the full
    30-layer 1.3B architecture (the C1 fixture's seeded weights, noise and prompts), FIFTY flow-match steps with CFG 5 — the loop of
    svi_video.py:392-421 with flow_match.py:53-64's update — once in fp32 and once the way the pipelines run it (bf16 weights,
    activations and latents).  Latents are kept after steps 1, 5, 10, 20, 30, 40 and 50 so the drift of bf16 Euler updates against the
    fp32 trajectory is a curve, not one number; the reference's own bf16-vs-fp32 gap at each of them is the yardstick."""
    import time
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    m = build_ref_dit(dit_mod, cfg, seed)
    noise = torch.randn((1, 16, 5, 32, 32), generator=torch.Generator("cpu").manual_seed(0), dtype=torch.float32)   # base.py:140-143
    pos = t(synth.text_context(seed + 1, 512, cfg["text_dim"], 64))
    neg = t(synth.text_context(seed + 2, 512, cfg["text_dim"], 64))
    keep = synth.C1_50_KEEP

    def loop(model, lat, pos, neg):
        s = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(50, shift=5.0)
        kept = []
        with torch.no_grad():
            for i, ts in enumerate(s.timesteps):
                tt = ts.unsqueeze(0)
                c = model(lat, tt, pos)
                u = model(lat, tt, neg)
                lat = s.step(u + 5.0 * (c - u), s.timesteps[i], lat)
                if i + 1 in keep:
                    kept.append(lat[0].float().numpy().copy())
        return np.stack(kept)

    t0 = time.time()
    l32 = loop(m, noise, pos, neg)
    print(f"c1_50step: fp32 loop {time.time() - t0:.0f} s")
    t0 = time.time()
    mb = m.to(torch.bfloat16)
    l16 = loop(mb, noise.to(torch.bfloat16), pos.to(torch.bfloat16), neg.to(torch.bfloat16))
    print(f"c1_50step: bf16 loop {time.time() - t0:.0f} s")
    gaps = [float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64))) for a, b in zip(l16, l32)]
    print("c1_50step: reference bf16-vs-fp32 rel-L2 after steps", dict(zip(keep, [f"{g:.3e}" for g in gaps])))
    np.savez(os.path.join(OUT, "c1_50step.npz"), steps=np.array(keep), latents_fp32=l32,
             latents_bf16_bits=synth.bf16_bits(l16), ref_gap=np.array(gaps))


def gen_dit_depth(dit_mod, fm):
    """The headline kernels at depth against the reference itself: the full 30-layer Wan2.1-T2V-1.3B architecture
    (the C1 fixture's seeded weights) on a (5,30,52) token grid = 7800 tokens — past the 2048-key threshold, so the HIP path takes the
    long-sequence attention kernel and the 256^2 GEMM in every block — through WanModel.forward (fp32 and, the way the pipelines run
    it, bf16) and a 2-step CFG-5 flow-match loop (svi_video.py:392-421).  fp32 results whole, bf16 results as 16-bit patterns."""
    import time
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    f, h, w = synth.DEPTH_GRID
    t0 = time.time()
    m = build_ref_dit(dit_mod, cfg, seed)
    print(f"depth: reference WanModel 1.3B built in {time.time() - t0:.0f} s")
    noise = torch.randn((1, 16, f, 2 * h, 2 * w), generator=torch.Generator("cpu").manual_seed(1), dtype=torch.float32)
    pos = t(synth.text_context(seed + 1, 512, cfg["text_dim"], 64))
    neg = t(synth.text_context(seed + 2, 512, cfg["text_dim"], 64))

    def loop(model, lat, pos, neg):
        s = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(synth.DEPTH_STEPS, shift=5.0)
        first = None
        with torch.no_grad():
            for i, ts in enumerate(s.timesteps):
                tt = ts.unsqueeze(0)
                c = model(lat, tt, pos)
                if first is None:
                    first = c
                u = model(lat, tt, neg)
                lat = s.step(u + 5.0 * (c - u), s.timesteps[i], lat)
        return first, lat

    t0 = time.time()
    f32, l32 = loop(m, noise, pos, neg)
    print(f"depth: fp32 loop {time.time() - t0:.0f} s")
    t0 = time.time()
    mb = m.to(torch.bfloat16)
    f16, l16 = loop(mb, noise.to(torch.bfloat16), pos.to(torch.bfloat16), neg.to(torch.bfloat16))
    print(f"depth: bf16 loop {time.time() - t0:.0f} s")
    np.savez(os.path.join(OUT, "dit_depth.npz"), fwd_fp32=f32[0].numpy(), lat_fp32=l32[0].numpy(),
             fwd_bf16_bits=synth.bf16_bits(f16[0].float().numpy()), lat_bf16_bits=synth.bf16_bits(l16[0].float().numpy()))


def gen_c4_blocks(dit_mod):
    """BASELINE configs[3]'s model end to end at a depth the host affords: WanModel with the Wan2.1-I2V-14B constructor table
    (wan_video_dit.py:699-712: dim 5120, 40 heads, ffn 13824, in_dim 36, image branch) cut to 4 of its 40 blocks, on the (3,20,36) grid =
    2160 tokens: the in_dim-36 patchify of x | y, img_emb on 257 CLIP tokens, four full blocks, head, unpatchify; fp32 and bf16."""
    import time
    cfg = dict(synth.WAN_14B_I2V, num_layers=synth.C4_LAYERS)
    seed = synth.C4_SEED
    f, h, w = synth.B14_GRID
    t0 = time.time()
    m = build_ref_dit(dit_mod, cfg, seed)
    print(f"c4: reference WanModel (14B-I2V widths, {synth.C4_LAYERS} blocks) built in {time.time() - t0:.0f} s")
    x = t(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    ctx = t(synth.text_context(seed + 2, 512, cfg["text_dim"], 64))
    clip = t(synth.randn(seed + 3, 1, 257, 1280))
    y = t(synth.randn(seed + 4, 1, 20, f, 2 * h, 2 * w))
    ts = torch.tensor([757.5758], dtype=torch.float32)
    with torch.no_grad():
        t0 = time.time()
        o32 = m(x, ts, ctx, clip_feature=clip, y=y)
        print(f"c4: fp32 forward {time.time() - t0:.0f} s")
        mb = m.to(torch.bfloat16)
        o16 = mb(x.to(torch.bfloat16), ts, ctx.to(torch.bfloat16), clip_feature=clip.to(torch.bfloat16), y=y.to(torch.bfloat16))
    np.savez(os.path.join(OUT, "dit_c4_4blocks.npz"), out_fp32=o32[0].numpy(), out_bf16_bits=synth.bf16_bits(o16[0].float().numpy()))


def gen_vae_c2(vae_mod):
    """The VAE at BASELINE config 2's spatial size (latent 60x104 <-> 480x832 px): the reference's decode of 2 latent frames
    (-> 5 frames) and encode of 5 frames (-> 2 latent frames).  The decoded video is stored on a stride-7 pixel lattice
    (7 is odd, so every pixel-tile alignment class of the kernels is hit); the latents whole."""
    v = vae_mod.WanVideoVAE()
    v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)
    k = synth.C2_VIDEO_STRIDE
    with torch.no_grad():
        z = t(synth.randn(511, 16, 2, 60, 104))
        video = v.decode([z], device="cpu")[0]                                         # [3,5,480,832]
        vid = t(np.tanh(synth.randn(512, 3, 5, 480, 832)))
        lat = v.encode([vid], device="cpu")[0]                                         # [16,2,60,104]
    np.savez(os.path.join(OUT, "vae_c2.npz"), decode_sample=video[:, :, ::k, ::k].contiguous().numpy(), encode=lat.numpy())


def gen_block_14b(dit_mod):
    """One DiTBlock at the Wan2.1-I2V-14B widths (dim 5120, 40 heads, ffn 13824, image branch with 257 CLIP tokens;
    wan_video_dit.py:699-712) on a (3,20,36) grid = 2160 tokens, fp32 and bf16, sampled rows."""
    cfg = dict(synth.WAN_14B_I2V, num_layers=1)
    seed, grid, nt = synth.B14_SEED, synth.B14_GRID, 512
    f, h, w = grid
    L = f * h * w
    sd = synth.dit_state_dict(seed, **cfg)
    blk = dit_mod.DiTBlock(True, cfg["dim"], cfg["dim"] // 128, cfg["ffn_dim"], 1e-6).eval()
    blk.load_state_dict({k[len("blocks.0."):]: t(a) for k, a in sd.items() if k.startswith("blocks.0.")}, strict=True)
    del sd
    bx = t(synth.randn(seed + 5, 1, L, cfg["dim"]))
    bctx = t(synth.randn(seed + 6, 1, nt + 257, cfg["dim"]))
    btm = t(0.5 * synth.randn(seed + 7, 1, 6, cfg["dim"]))
    fr = dit_mod.precompute_freqs_cis_3d(128)
    freqs = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                       fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
    rows = synth.B14_ROWS(L)
    with torch.no_grad():
        o32 = blk(bx, bctx, btm, freqs)[0, rows].numpy()
        blk = blk.to(torch.bfloat16)
        o16 = blk(bx.to(torch.bfloat16), bctx.to(torch.bfloat16), btm.to(torch.bfloat16), freqs)[0, rows].float().numpy()
    np.savez(os.path.join(OUT, "dit_block_14b.npz"), block_fp32=o32, block_bf16=o16, rows=np.asarray(rows))


def _ref_block_case(dit_mod, cfg, seed, grid, has_img, fname, row_stride=244):
    """ONE DiTBlock.forward of the reference (wan_video_dit.py:354-374) at the full C2 token count, fp32 and bf16, C2_ROWS kept."""
    import time
    f, h, w = grid
    L = f * h * w
    nt = 512
    sd = synth.dit_state_dict(seed, **cfg)
    blk = dit_mod.DiTBlock(has_img, cfg["dim"], cfg["dim"] // 128, cfg["ffn_dim"], 1e-6).eval()
    blk.load_state_dict({k[len("blocks.0."):]: t(a) for k, a in sd.items() if k.startswith("blocks.0.")}, strict=True)
    del sd
    bx = t(synth.randn(seed + 5, 1, L, cfg["dim"]))
    bctx = t(synth.randn(seed + 6, 1, nt + (257 if has_img else 0), cfg["dim"]))
    bctx[:, (257 if has_img else 0) + 64:] = 0          # zero-padded prompt (wan_prompter.py:107-108): 64 live text tokens of 512
    btm = t(0.5 * synth.randn(seed + 7, 1, 6, cfg["dim"]))
    fr = dit_mod.precompute_freqs_cis_3d(128)
    freqs = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                       fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(L, 1, -1)
    rows = synth.C2_ROWS(L, row_stride)
    with torch.no_grad():
        t0 = time.time()
        o32 = blk(bx, bctx, btm, freqs)[0, rows].numpy()
        print(f"{fname}: reference DiTBlock fp32 at L={L}: {time.time() - t0:.0f} s", flush=True)
        t0 = time.time()
        blk = blk.to(torch.bfloat16)
        o16 = blk(bx.to(torch.bfloat16), bctx.to(torch.bfloat16), btm.to(torch.bfloat16), freqs)[0, rows].float().numpy()
        print(f"{fname}: bf16: {time.time() - t0:.0f} s", flush=True)
    np.savez(os.path.join(OUT, fname), block_fp32=o32, block_bf16_bits=synth.bf16_bits(o16), rows=np.asarray(rows))


def gen_block_c2(dit_mod):
    """The reference's own DiTBlock at the headline size: Wan2.1-T2V-1.3B widths, (21,30,52) grid = 32760 tokens."""
    _ref_block_case(dit_mod, dict(synth.WAN_1_3B, num_layers=1), synth.B13C2_SEED, synth.C2_GRID, False, "dit_block_c2.npz")


def gen_block_14b_c2(dit_mod):
    """C4 at size: one Wan2.1-I2V-14B DiTBlock (dim 5120, 40 heads, ffn 13824, 257 CLIP + 512 text context) at 32760 tokens."""
    _ref_block_case(dit_mod, dict(synth.WAN_14B_I2V, num_layers=1), synth.B14C2_SEED, synth.C2_GRID, True, "dit_block_14b_c2.npz", row_stride=488)


def gen_c2_full(dit_mod):
    """Size x depth together: the reference's 30-layer Wan2.1-T2V-1.3B WanModel.forward (wan_video_dit.py:486-567) on the
    full C2 latent [1,16,21,60,104] = 32760 tokens — the configuration bench.py times — in fp32 and, the way the pipelines run it, bf16.
    C1_SEED weights, torch CPU-generator noise (base.py:140-143), zero-padded 512-token prompt.  Output kept on a stride-3 (h, w) lattice."""
    import time
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    f, h, w = synth.C2_GRID
    k = synth.C2_FULL_STRIDE
    t0 = time.time()
    m = build_ref_dit(dit_mod, cfg, seed)
    print(f"c2_full: reference WanModel 1.3B built in {time.time() - t0:.0f} s", flush=True)
    noise = torch.randn((1, 16, f, 2 * h, 2 * w), generator=torch.Generator("cpu").manual_seed(2), dtype=torch.float32)
    pos = t(synth.text_context(seed + 1, 512, cfg["text_dim"], 64))
    ts = torch.tensor([991.7355], dtype=torch.float32)
    with torch.no_grad():
        t0 = time.time()
        o32 = m(noise, ts, pos)[0]
        print(f"c2_full: fp32 forward {time.time() - t0:.0f} s", flush=True)
        np.savez(os.path.join(OUT, "dit_c2_full.npz"), out_fp32=o32[:, :, ::k, ::k].contiguous().numpy())          # kept in case the bf16 pass dies
        t0 = time.time()
        mb = m.to(torch.bfloat16)
        o16 = mb(noise.to(torch.bfloat16), ts, pos.to(torch.bfloat16))[0].float()
        print(f"c2_full: bf16 forward {time.time() - t0:.0f} s", flush=True)
    np.savez(os.path.join(OUT, "dit_c2_full.npz"), out_fp32=o32[:, :, ::k, ::k].contiguous().numpy(),
             out_bf16_bits=synth.bf16_bits(o16[:, :, ::k, ::k].contiguous().numpy()),
             out_fp32_norm=np.float64(o32.double().norm().item()), full_rel_bf16_vs_fp32=np.float64(((o16 - o32).double().norm() / o32.double().norm()).item()))


def gen_c2_loop(dit_mod, fm):
    """The LOOP at the headline size: a 2-step CFG-5 flow-match loop (svi_video.py:392-421: cond forward, uncond forward,
    u + 5 (c - u), scheduler.step) of the reference's 30-layer 1.3B WanModel on the full C2 latent [1,16,21,60,104] = 32760 tokens: 4 forwards
    in fp32 and 4 in bf16 (the way the pipelines run it).  Weights, noise and the positive prompt are dit_c2_full.npz's; the negative prompt has
    32 valid rows (so the two branches walk different key counts).  Final latents kept on the stride-3 (h, w) lattice."""
    import time
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    f, h, w = synth.C2_GRID
    k = synth.C2_FULL_STRIDE
    t0 = time.time()
    m = build_ref_dit(dit_mod, cfg, seed)
    print(f"c2_loop: reference WanModel 1.3B built in {time.time() - t0:.0f} s", flush=True)
    noise = torch.randn((1, 16, f, 2 * h, 2 * w), generator=torch.Generator("cpu").manual_seed(2), dtype=torch.float32)
    pos = t(synth.text_context(seed + 1, 512, cfg["text_dim"], 64))
    neg = t(synth.text_context(seed + 2, 512, cfg["text_dim"], 32))

    def loop(model, lat, pos, neg, tag):
        s = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(synth.C2_LOOP_STEPS, shift=5.0)
        with torch.no_grad():
            for i, ts in enumerate(s.timesteps):
                tt = ts.unsqueeze(0)
                t1 = time.time()
                c = model(lat, tt, pos)
                u = model(lat, tt, neg)
                lat = s.step(u + 5.0 * (c - u), s.timesteps[i], lat)
                print(f"c2_loop: {tag} step {i} {time.time() - t1:.0f} s", flush=True)
        return lat

    l32 = loop(m, noise, pos, neg, "fp32")[0]
    np.savez(os.path.join(OUT, "dit_c2_loop.npz"), lat_fp32=l32[:, :, ::k, ::k].contiguous().numpy())          # kept in case the bf16 pass dies
    mb = m.to(torch.bfloat16)
    l16 = loop(mb, noise.to(torch.bfloat16), pos.to(torch.bfloat16), neg.to(torch.bfloat16), "bf16")[0].float()
    np.savez(os.path.join(OUT, "dit_c2_loop.npz"), lat_fp32=l32[:, :, ::k, ::k].contiguous().numpy(),
             lat_bf16_bits=synth.bf16_bits(l16[:, :, ::k, ::k].contiguous().numpy()),
             lat_fp32_norm=np.float64(l32.double().norm().item()), full_rel_bf16_vs_fp32=np.float64(((l16 - l32).double().norm() / l32.double().norm()).item()))


def gen_vae_c2_full(vae_mod):
    """The whole 81-frame decode at the headline size against the reference (frames 6..81 were pinned by causality only):
    WanVideoVAE.decode (wan_video_vae.py:777-789) of the 21-latent-frame tensor tests/test_gpu_configs.py builds (seeds 511 | 513); kept:
    frames 40..44 and 76..80 on the stride-7 pixel lattice."""
    import time
    v = vae_mod.WanVideoVAE()
    v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)
    k = synth.C2_VIDEO_STRIDE
    z21 = torch.cat([t(synth.randn(511, 16, 2, 60, 104)), t(synth.randn(513, 16, 19, 60, 104))], dim=1)
    t0 = time.time()
    with torch.no_grad():
        video = v.decode([z21], device="cpu")[0]                                       # [3,81,480,832]
    print(f"vae_c2_full: reference decode of 21 latent frames {time.time() - t0:.0f} s", flush=True)
    assert tuple(video.shape) == (3, 81, 480, 832)
    np.savez(os.path.join(OUT, "vae_c2_full.npz"), mid=video[:, 40:45, ::k, ::k].contiguous().numpy(), tail=video[:, 76:81, ::k, ::k].contiguous().numpy())


def gen_vae_tiled(vae_mod):
    """WanVideoVAE.tiled_decode / tiled_encode (wan_video_vae.py:643-744) on multi-tile problems, through the public
    decode/encode(tiled=True): 3x3 and ragged tile grids, including tile values beyond +-1 before the blend's final clamp."""
    v = vae_mod.WanVideoVAE()
    v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)
    out = {}
    with torch.no_grad():
        for name, zshape, size, stride, seed in synth.TILED_DECODE_CASES:
            z = t(2.0 * synth.randn(seed, *zshape))
            out["decode_" + name] = v.decode([z], device="cpu", tiled=True, tile_size=size, tile_stride=stride)[0].numpy()
        for name, vshape, size, stride, seed in synth.TILED_ENCODE_CASES:
            vid = t(np.tanh(synth.randn(seed, *vshape)))
            out["encode_" + name] = v.encode([vid], device="cpu", tiled=True, tile_size=size, tile_stride=stride)[0].numpy()
            # a batch of two: the reference rescales tile_size inside its per-video loop (vae:765-767) -> the second video sees 8x larger tiles
            out["encode_" + name + "_batch_second"] = v.encode([vid, vid], device="cpu", tiled=True, tile_size=size, tile_stride=stride)[1].numpy()
    np.savez(os.path.join(OUT, "vae_tiled.npz"), **out)


def _reference_clip_loop(namespace):
    """The clip loop of the reference's inference script — the `for chunk_idx in range(num_clips):` statement of the script body
    (test_svi.py:424-485) — compiled out of the script and executed in `namespace` (which supplies pipe, args, prompts, ...)."""
    import ast
    tree = ast.parse(open(os.path.join(REF, "test_svi.py")).read())
    loop = next(n for n in ast.walk(tree) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == "chunk_idx")
    exec(compile(ast.Module(body=[loop], type_ignores=[]), os.path.join(REF, "test_svi.py"), "exec"), namespace)
    return namespace


def gen_clip_stream(dit_mod, vae_mod, fm):
    """Rows a23 / b: the reference's OWN clip loop (test_svi.py:424-485) around the reference's OWN SVIVideoPipeline.__call__
    (svi_video.py:423-520, with encode_images_adaptive, _sample_with_regular_video, decode_video, tensor2video and the
    BasePipeline helpers), every one of them compiled out of its source file and run on a stand-in pipeline object that carries
    the reference WanModel (tiny I2V config, bf16 as the pipelines run it), the reference WanVideoVAE (fp32) and the reference
    FlowMatchScheduler.  Outside SURVEY §8 and therefore injected: the T5 prompt embeddings (a table prompt -> seeded tensor)
    and the CLIP image feature (a seeded constant).  Stored: per clip the call's arguments (seed, prompt, motion frames), the
    8-bit frames it returned, the stitched video, and clip 0's float video before tensor2video."""
    import contextlib
    import io
    import types as _types
    from PIL import Image
    from einops import rearrange
    c = synth.TINY_DIT_I2V
    ns = {"torch": torch, "np": np, "Image": Image, "rearrange": rearrange, "tqdm": lambda x, **k: x, "Optional": None, "os": os,
          "WanModel": dit_mod.WanModel, "sinusoidal_embedding_1d": dit_mod.sinusoidal_embedding_1d}
    ns["TeaCache"] = _reference_toplevel("diffsynth/pipelines/svi_video.py", "TeaCache", ns)
    ns["model_fn_wan_video"] = _reference_toplevel("diffsynth/pipelines/svi_video.py", "model_fn_wan_video", ns)
    members = {}
    for name in ("__call__", "encode_images_adaptive", "tensor2video", "prepare_extra_input", "decode_video",
                 "_sample_with_regular_video", "prepare_unified_sequence_parallel"):
        members[name] = _reference_method("diffsynth/pipelines/svi_video.py", "SVIVideoPipeline", name, ns)
    for name in ("check_resize_height_width", "preprocess_image", "generate_noise"):
        members[name] = _reference_method("diffsynth/pipelines/base.py", "BasePipeline", name, ns)
    Pipe = type("StandInPipeline", (), members)

    class Clip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def encode_image(self, images):
            return t(synth.randn(synth.STREAM_CLIP_SEED, 1, 257, 1280))

    out = {}
    for case in synth.STREAM_CASES:
        name, n_motion, num_frames, num_clips, steps = case["name"], case["num_motion_frames"], case["num_frames"], case["num_clips"], case["steps"]
        H, W = synth.STREAM_HW
        pipe = Pipe()
        pipe.torch_dtype, pipe.device = torch.bfloat16, "cpu"
        pipe.height_division_factor = pipe.width_division_factor = 16                # SVIVideoPipeline.__init__, svi_video.py:151-152
        pipe.use_unified_sequence_parallel = False
        pipe.scheduler = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True, num_train_timesteps=1000)   # :144
        m = dit_mod.WanModel(eps=1e-6, num_heads=synth.num_heads_of(c), **c)
        m.load_state_dict({k: t(a) for k, a in synth.dit_state_dict(200, **c).items()}, strict=True)
        pipe.dit = m.to(torch.bfloat16).eval()
        v = vae_mod.WanVideoVAE()
        v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)
        pipe.vae, pipe.image_encoder = v, Clip()
        pipe.load_models_to_device = lambda names=[]: None
        prompts = [f"prompt {i}" for i in range(case["num_prompts"])]
        table = {p: t(synth.text_context(synth.STREAM_PROMPT_SEED + i, 16, c["text_dim"], 9)).to(torch.bfloat16) for i, p in enumerate(prompts)}
        table["negative"] = t(synth.text_context(synth.STREAM_PROMPT_SEED + 50, 16, c["text_dim"], 5)).to(torch.bfloat16)
        pipe.encode_prompt = lambda prompt, positive=True: {"context": table[prompt]}
        calls, floats = [], []
        real_t2v = pipe.tensor2video
        pipe.tensor2video = lambda frames: (floats.append(frames.float().clone()), real_t2v(frames))[1]

        def call(**kw):
            img = kw["input_image"]
            img = img if isinstance(img, list) else [img]
            frames = pipe(num_frames=num_frames, **kw)                               # the loop leaves num_frames at its default (81)
            calls.append(dict(seed=kw["seed"], prompt=prompts.index(kw["prompt"]), motion=np.stack([np.array(i) for i in img]),
                              frames=np.stack([np.array(f) for f in frames])))
            return frames

        first = Image.fromarray(synth.condition_frames(synth.STREAM_IMAGE_SEED, 1, H, W)[0])
        args = _types.SimpleNamespace(seed_times=42, use_first_prompt_only=case["use_first_prompt_only"], prompt_repeat_times=case["prompt_repeat_times"],
                                      prompt_prefix="none", num_steps=steps, cfg_scale_text=5.0, tiled=False, ref_pad_cfg=case["ref_pad_cfg"],
                                      ref_pad_num=case["ref_pad_num"])
        loop_ns = dict(num_clips=num_clips, seeds=range(0, 10000), args=args, loaded_prompts=prompts, path_dir_per={"negative_prompt": "negative", "prompt_name": name},
                       pipe=call, rand_ref_frame_final=first, rand_ref_frame_final_gt=torch.from_numpy(np.array(first)).clone(), height=H, width=W,
                       use_teacache=False, num_motion_frames=n_motion, video_list=[], sample_output_dir="", base_filename="", os=os,
                       save_video=lambda *a, **k: None, ref_name=name)
        with contextlib.redirect_stdout(io.StringIO()):
            _reference_clip_loop(loop_ns)
        stitched = np.stack([np.array(f) for f in loop_ns["video_list"]])
        out[name + "_stitched"] = stitched
        out[name + "_frames"] = np.stack([cl["frames"] for cl in calls])
        out[name + "_seeds"] = np.array([cl["seed"] for cl in calls])
        out[name + "_prompts"] = np.array([cl["prompt"] for cl in calls])
        for k, cl in enumerate(calls):
            out[f"{name}_motion{k}"] = cl["motion"]
        out[name + "_video_f32_clip0"] = floats[0].numpy()
        print("stream", name, "stitched", stitched.shape, "seeds", out[name + "_seeds"], "prompts", out[name + "_prompts"])
    np.savez_compressed(os.path.join(OUT, "clip_stream.npz"), **out)


def gen_pose_embed():
    """Row N3: the dance variant's pose embedder.  The nn.Sequential is built by evaluating the reference's OWN expression (the value
    assigned to self.dwpose_embedding in SVIDanceVideoPipeline.fetch_models, svi_video_dance.py:255-269) and driven by the reference's
    OWN statements (the body of `if humanpose_data is not None:` in __call__, :527-530), both compiled out of the source file."""
    import ast
    from einops import rearrange
    import torch.nn as nn
    path = os.path.join(REF, "diffsynth/pipelines/svi_video_dance.py")
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SVIDanceVideoPipeline")
    fetch = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "fetch_models")
    assign = next(n for n in ast.walk(fetch) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Attribute)
                  and n.targets[0].attr == "dwpose_embedding")
    seq = eval(compile(ast.Expression(assign.value), path, "eval"), {"nn": nn, "concat_dim": 4})
    seq.load_state_dict({k: t(a) for k, a in synth.pose_state_dict(synth.POSE_SEED).items()}, strict=True)
    call = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__call__")
    branch = next(n for n in ast.walk(call) if isinstance(n, ast.If) and isinstance(n.test, ast.Compare)
                  and isinstance(n.test.left, ast.Name) and n.test.left.id == "humanpose_data")
    code = compile(ast.Module(body=branch.body, type_ignores=[]), path, "exec")

    class Self:
        device = "cpu"
        dwpose_embedding = seq

    out = {}
    with torch.no_grad():
        for name, shape, seed in synth.POSE_CASES:
            ns = {"self": Self(), "torch": torch, "rearrange": rearrange, "humanpose_data": t(synth.pose_video(seed, *shape)),
                  "latents": torch.zeros(1, dtype=torch.bfloat16)}
            exec(code, ns)
            out[name] = ns["condition"].float().numpy()
            print("pose", name, out[name].shape)
    np.savez(os.path.join(OUT, "pose_embed.npz"), **out)


def gen_dance_sampler(dit_mod, fm):
    """The dance variant's sampler: SVIDanceVideoPipeline._sample_with_dance_video and that module's own model_fn_wan_video
    (svi_video_dance.py:414-443, :74-137), compiled out of the source file, on the reference WanModel (tiny I2V config, bf16 as the
    pipelines run it): `add_condition` reaches the conditional forward only — unless cond_wo_pose."""
    rel = "diffsynth/pipelines/svi_video_dance.py"
    ns = {"torch": torch, "np": np, "WanModel": dit_mod.WanModel, "Optional": None, "sinusoidal_embedding_1d": dit_mod.sinusoidal_embedding_1d}
    ns["TeaCache"] = _reference_toplevel(rel, "TeaCache", ns)
    ns["model_fn_wan_video"] = _reference_toplevel(rel, "model_fn_wan_video", ns)
    sample = _reference_method(rel, "SVIDanceVideoPipeline", "_sample_with_dance_video", ns)
    c, seed, grid = synth.TINY_DIT_I2V, 200, (2, 4, 4)
    f, h, w = grid
    m = dit_mod.WanModel(eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    m.load_state_dict({k: t(a) for k, a in synth.dit_state_dict(seed, **c).items()}, strict=True)

    class Self:
        device = "cpu"

    me = Self()
    me.dit = m.to(torch.bfloat16).eval()
    out = {}
    bf = lambda a: t(a).to(torch.bfloat16)      # noqa: E731
    for name, wo in (("cond_only", False), ("cond_wo_pose", True)):
        me.scheduler = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        me.scheduler.set_timesteps(3, shift=5.0)
        lat = torch.randn((1, 16, f, 2 * h, 2 * w), generator=torch.Generator("cpu").manual_seed(21), dtype=torch.float32).to(torch.bfloat16)
        image_emb = {"clip_feature": bf(synth.randn(seed + 3, 1, 257, 1280)), "y": bf(synth.randn(seed + 4, 1, 20, f, 2 * h, 2 * w))}
        with torch.no_grad():
            r = sample(me, lat, {"context": bf(synth.text_context(seed + 2, 16, c["text_dim"], 10))},
                       {"context": bf(synth.text_context(seed + 12, 16, c["text_dim"], 4))}, image_emb, {}, {"tea_cache": None}, {"tea_cache": None},
                       {"use_unified_sequence_parallel": False}, False, {"text": 5.0}, lambda x: x,
                       add_condition=bf(0.5 * synth.randn(seed + 9, 1, f * h * w, c["dim"])), cond_wo_pose=wo)
        out[name] = r.float().numpy()
    np.savez(os.path.join(OUT, "dance_sampler.npz"), **out)


def gen_fp8_storage(dit_mod):
    """SURVEY F4/F5 (BASELINE configs[4], north_star "bf16/fp8"): the reference's FP8 mode stores every parameter as float8_e4m3fn
    (test_svi.py:337: load_models(torch_dtype=torch.float8_e4m3fn)) and casts it to the bf16 computation dtype in front of every use
    (vram_management/layers.py:65-71, cast_to = weight.to(dtype)).  Its result is therefore the bf16 forward of the reference model on
    weights bf16(e4m3(W)): exactly that, on the tiny T2V config."""
    c, grid, nt, nv, ts, seed = synth.TINY_DIT, (3, 4, 6), 20, 13, 637.5, 100
    f, h, w = grid
    m = build_ref_dit(dit_mod, c, seed)
    with torch.no_grad():
        for p in m.parameters():
            p.data = p.data.to(torch.float8_e4m3fn).to(torch.bfloat16)
        m = m.to(torch.bfloat16).eval()
        x = t(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w)).to(torch.bfloat16)
        ctx = t(synth.text_context(seed + 2, nt, c["text_dim"], nv)).to(torch.bfloat16)
        out = m(x, torch.tensor([ts]), ctx).float().numpy()
    # every e4m3fn code point through torch's own cast, as the known-answer table of the decode
    codes = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).to(torch.bfloat16).view(torch.int16).numpy()
    np.savez(os.path.join(OUT, "fp8_storage.npz"), out_bf16=out, e4m3_to_bf16_bits=codes)


def gen_lora_names():
    """GeneralLoRAFromPeft.get_name_dict (models/lora.py:204-217), compiled out of the source file, on the key styles SVI's LoRA files use."""
    import json
    fn = _reference_method("diffsynth/models/lora.py", "GeneralLoRAFromPeft", "get_name_dict", {})
    keys = synth.LORA_KEY_EXAMPLES
    got = fn(None, {k: None for k in keys})
    with open(os.path.join(OUT, "lora_names.json"), "w") as f:
        json.dump({k: list(v) for k, v in got.items()}, f, indent=1, sort_keys=True)


def _sdpa_as_xformers(q, k, v, attn_bias=None, op=None):
    """xformers.ops.memory_efficient_attention(q, k, v) on [B, M, H, K] operands, stated with torch's SDPA (xformers is absent here;
    the reference treats its attention back ends as interchangeable)."""
    assert attn_bias is None
    o = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return o.transpose(1, 2)


def _talk_model(dit_mod):
    xf = sys.modules["xformers"]                                   # the import stub (import_reference); SingleStreamAttention calls
    xf.ops = sys.modules["xformers.ops"]                           # xformers.ops.memory_efficient_attention (models/attention.py:357)
    xf.ops.memory_efficient_attention = _sdpa_as_xformers
    c = synth.TINY_DIT_TALK
    m = dit_mod.WanModel(eps=1e-6, num_heads=synth.num_heads_of(c), **c).eval()
    m.load_state_dict({k: t(a) for k, a in synth.dit_state_dict(synth.TALK_SEED, **c).items()}, strict=True)
    return m, c


def gen_talk(dit_mod, fm):
    """Talk variant: the reference WanModel with enable_multitalk (per-block audio cross-attention, AudioProjModel) run on
    seeded audio windows — WanModel.forward(audio_embed_tuple=...) (wan_video_dit.py:486-567) in fp32 and bf16 — and the three-way-guidance
    sampler SVITalkVideoPipeline._sample_with_multitalk with that module's own model_fn_wan_talk_video (svi_video_talk.py:83-160,448-463)."""
    m, c = _talk_model(dit_mod)
    seed, (f, h, w) = synth.TALK_SEED, synth.TALK_GRID
    x = t(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    ctx = t(synth.text_context(seed + 2, 20, c["text_dim"], 13))
    clip = t(synth.randn(seed + 3, 1, 257, 1280))
    y = t(synth.randn(seed + 4, 1, 20, f, 2 * h, 2 * w))
    a0, a1 = (t(a) for a in synth.audio_windows(seed + 5, f))
    ts = torch.tensor([637.5])
    out = {}
    with torch.no_grad():
        out["out_fp32"] = m(x, ts, ctx, clip_feature=clip, y=y, audio_embed_tuple=(a0, a1)).numpy()
        out["out_fp32_no_audio"] = m(x, ts, ctx, clip_feature=clip, y=y).numpy()
        out["audio_tokens_fp32"] = m.audio_proj(a0, a1)[0].numpy()                    # [f, 32, 768]
        mb = m.to(torch.bfloat16)
        bf = lambda a: a.to(torch.bfloat16)      # noqa: E731
        out["out_bf16"] = mb(bf(x), ts, bf(ctx), clip_feature=bf(clip), y=bf(y), audio_embed_tuple=(bf(a0), bf(a1))).float().numpy()
        # the sampler (3 steps): cond / uncond / drop-text forwards, text scale 5, audio scale 4
        rel = "diffsynth/pipelines/svi_video_talk.py"
        ns = {"torch": torch, "np": np, "WanModel": dit_mod.WanModel, "Optional": None, "sinusoidal_embedding_1d": dit_mod.sinusoidal_embedding_1d}
        ns["TeaCache"] = _reference_toplevel(rel, "TeaCache", ns)
        ns["model_fn_wan_talk_video"] = _reference_toplevel(rel, "model_fn_wan_talk_video", ns)
        sample = _reference_method(rel, "SVITalkVideoPipeline", "_sample_with_multitalk", ns)

        class Self:
            device = "cpu"

        me = Self()
        me.dit = mb
        me.scheduler = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        me.scheduler.set_timesteps(3, shift=5.0)
        lat = torch.randn((1, 16, f, 2 * h, 2 * w), generator=torch.Generator("cpu").manual_seed(31), dtype=torch.float32).to(torch.bfloat16)
        n0, n1 = (bf(t(a)) for a in synth.audio_windows(seed + 7, f))                 # the "null" audio of the unconditional branch
        r = sample(me, lat, {"context": bf(ctx)}, {"context": bf(t(synth.text_context(seed + 12, 20, c["text_dim"], 4)))},
                   {"clip_feature": bf(clip), "y": bf(y)}, {}, {"tea_cache": None}, {"tea_cache": None}, {"use_unified_sequence_parallel": False},
                   None, (bf(a0), bf(a1)), (n0, n1), False, {"text": 5.0, "audio": 4.0}, lambda x: x)
        out["sampler_latents"] = r.float().numpy()
    np.savez(os.path.join(OUT, "dit_tiny_talk.npz"), **out)


def gen_t5():
    """Row N4: the reference's own WanTextEncoder (models/wan_video_text_encoder.py) on seeded weights, in fp32 and as the bf16 module the
    pipeline keeps, driven through WanPrompter.encode_prompt (prompters/wan_prompter.py:99-112, compiled out of the source file) with a
    stand-in tokenizer that hands over seeded (ids, mask); plus the relative-position bucket table of T5RelativeEmbedding."""
    te = importlib.import_module("diffsynth.models.wan_video_text_encoder")
    encode_prompt = _reference_method("diffsynth/prompters/wan_prompter.py", "WanPrompter", "encode_prompt", {"torch": torch})
    out = {}
    rel = torch.arange(512).unsqueeze(0) - torch.arange(512).unsqueeze(1)          # key - query, as T5RelativeEmbedding.forward builds it
    buckets = te.T5RelativeEmbedding(32, 64, bidirectional=True)._relative_position_bucket(rel)
    out["buckets_512"] = np.concatenate([buckets[-1, :511].numpy()[::1], buckets[0].numpy()]).astype(np.int32)     # rel = -511..-1, 0..511
    assert all(int(buckets[i, j]) == int(out["buckets_512"][j - i + 511]) for i, j in ((0, 0), (5, 300), (300, 5), (511, 0), (0, 511), (100, 132)))

    def run(cfg, seed, cases, rows=None):
        sd = {k: t(a) for k, a in synth.t5_state_dict(seed, **cfg).items()}
        for dtype, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
            m = te.WanTextEncoder(**cfg).eval()
            m.load_state_dict(sd, strict=True)
            m = m.to(dtype)

            class Prompter:
                text_encoder = m

                def process_prompt(self, prompt, positive=True):
                    return prompt

            for name, L, valid, cseed in cases:
                ids, mask = synth.t5_ids(cseed, L, valid, cfg["vocab"])
                me = Prompter()
                me.tokenizer = lambda prompt, return_mask, add_special_tokens: (t(ids), t(mask))
                with torch.no_grad():
                    full = m(t(ids), t(mask)).float().numpy()[0]
                    emb = encode_prompt(me, "a prompt", positive=True, device="cpu").float().numpy()[0]
                assert np.array_equal(emb[:valid], full[:valid]) and not emb[valid:].any()
                out[f"{name}_{tag}"] = full if rows is None else full[rows]
                print("t5", name, tag, full.shape, float(np.abs(full).max()))

    run(synth.T5_TINY, synth.T5_SEED, synth.T5_TINY_CASES)
    run(synth.T5_XXL_BLOCK, synth.T5_SEED + 1, [synth.T5_XXL_CASE], synth.T5_XXL_ROWS)
    run(synth.T5_DEEP, synth.T5_SEED + 2, [synth.T5_DEEP_CASE])
    np.savez(os.path.join(OUT, "t5_encoder.npz"), **out)


def gen_clip():
    """Row N4: WanImageEncoder.encode_image (models/wan_video_image_encoder.py:864-880), compiled out of the source file, around the
    reference's own VisionTransformer built the way XLMRobertaCLIP builds it (:686-701), fp32 as SVI runs it.  torchvision is absent in
    the build container: `self.transforms.transforms[-1]` is stated as what torchvision.transforms.Normalize does, (x - mean) / std
    per channel with the CLIP constants of :783-784."""
    ie = importlib.import_module("diffsynth.models.wan_video_image_encoder")
    import torch.nn.functional as F
    encode_image = _reference_method("diffsynth/models/wan_video_image_encoder.py", "WanImageEncoder", "encode_image", {"torch": torch, "F": F})
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    out = {}

    def run(cfg, seed, cases, rows=None):
        vis = ie.VisionTransformer(image_size=cfg["image_size"], patch_size=cfg["patch_size"], dim=cfg["dim"], mlp_ratio=cfg["mlp_ratio"],
                                   out_dim=1024 if cfg["dim"] == 1280 else 64, num_heads=cfg["num_heads"], num_layers=cfg["num_layers"],
                                   pool_type="token", pre_norm=True, post_norm=False, activation="gelu", norm_eps=1e-5).eval()
        vis.load_state_dict({k: t(a) for k, a in synth.clip_state_dict(seed, **cfg).items()}, strict=True)

        class Model:
            image_size = cfg["image_size"]
            visual = vis

        class Self:
            model = Model()
            transforms = types.SimpleNamespace(transforms=[lambda x: x.sub_(mean).div_(std)])

        for name, shape, cseed in cases:
            img = synth.clip_image(cseed, *shape)
            with torch.no_grad():
                o = encode_image(Self(), [t(img.copy())]).numpy()
            out[name] = o if rows is None else o[:, rows]
            print("clip", name, o.shape, float(np.abs(o).max()))

    run(synth.CLIP_TINY, synth.CLIP_SEED, synth.CLIP_TINY_CASES)
    run(synth.CLIP_H_BLOCK, synth.CLIP_SEED + 1, [synth.CLIP_H_CASE], synth.CLIP_H_ROWS)
    run(synth.CLIP_DEEP, synth.CLIP_SEED + 2, [synth.CLIP_DEEP_CASE])
    np.savez(os.path.join(OUT, "clip_encoder.npz"), **out)


def main(argv=None):
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    dit_mod, vae_mod, fm = import_reference()
    jobs = {
        "flow_match": lambda: gen_flow_match(fm),
        "dit_tiny_t2v": lambda: dit_case(dit_mod, "tiny_t2v", synth.TINY_DIT, (3, 4, 6), 20, 13, 637.5, 100),
        "dit_small_t2v": lambda: dit_case(dit_mod, "small_t2v", synth.SMALL_DIT, (2, 5, 7), 24, 24, 991.7355, 150),
        "dit_tiny_i2v": lambda: dit_case(dit_mod, "tiny_i2v", synth.TINY_DIT_I2V, (2, 4, 4), 16, 10, 92.5926, 200),
        "denoise_tiny": lambda: gen_denoise(dit_mod, fm),
        "vae": lambda: gen_vae(vae_mod),
        "image_condition": lambda: gen_image_condition(vae_mod),
        "teacache_tiny": lambda: gen_teacache(dit_mod, fm),
        "vae_tiled": lambda: gen_vae_tiled(vae_mod),
        "vae_c2": lambda: gen_vae_c2(vae_mod),
        "dit_block_14b": lambda: gen_block_14b(dit_mod),
        "clip_stream": lambda: gen_clip_stream(dit_mod, vae_mod, fm),
        "fp8_storage": lambda: gen_fp8_storage(dit_mod),
        "lora_names": gen_lora_names,
        "dit_tiny_talk": lambda: gen_talk(dit_mod, fm),
        "pose_embed": gen_pose_embed,
        "dance_sampler": lambda: gen_dance_sampler(dit_mod, fm),
        "t5_encoder": gen_t5,
        "clip_encoder": gen_clip,
        "c1_e2e": lambda: gen_c1_e2e(dit_mod, vae_mod, fm),
        "c1_50step": lambda: gen_c1_50step(dit_mod, fm),
        "dit_c4_4blocks": lambda: gen_c4_blocks(dit_mod),
        "dit_depth": lambda: gen_dit_depth(dit_mod, fm),
        "dit_block_c2": lambda: gen_block_c2(dit_mod),
        "dit_block_14b_c2": lambda: gen_block_14b_c2(dit_mod),
        "dit_c2_full": lambda: gen_c2_full(dit_mod),
        "dit_c2_loop": lambda: gen_c2_loop(dit_mod, fm),
        "vae_c2_full": lambda: gen_vae_c2_full(vae_mod),
    }
    names = list(argv if argv is not None else sys.argv[1:]) or list(jobs)
    for n in names:
        jobs[n]()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
