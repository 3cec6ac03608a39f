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
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m

    for pkg in ("diffsynth", "diffsynth.models", "diffsynth.utils", "diffsynth.schedulers"):
        ns(pkg, REF + "/" + pkg.replace(".", "/"))

    class Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (object,), {})

    for n in ("diffusers", "diffusers.configuration_utils", "xfuser", "xfuser.core", "xfuser.core.distributed",
              "xformers", "xformers.ops", "imageio", "torchvision"):
        sys.modules.setdefault(n, Stub(n))
    sys.modules["diffusers.configuration_utils"].register_to_config = lambda f: f
    dit = importlib.import_module("diffsynth.models.wan_video_dit")
    vae = importlib.import_module("diffsynth.models.wan_video_vae")
    fm = importlib.import_module("diffsynth.schedulers.flow_match")
    return dit, vae, fm


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
    reference VAE (seeded weights), BasePipeline.preprocess_image and a CLIP stub: pins row a22 (y = mask | VAE latent)."""
    from PIL import Image
    ns = {"torch": torch, "np": np}
    encode_images_adaptive = _reference_method("diffsynth/pipelines/svi_video.py", "SVIVideoPipeline", "encode_images_adaptive", ns)
    preprocess_image = _reference_method("diffsynth/pipelines/base.py", "BasePipeline", "preprocess_image", ns)
    v = vae_mod.WanVideoVAE()
    v.load_state_dict({k: t(a) for k, a in synth.vae_state_dict(500).items()}, strict=True)

    class Clip(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def encode_image(self, images):
            return torch.zeros(1, 257, 1280)

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
            out[name] = r["y"].float().numpy()
    np.savez(os.path.join(OUT, "image_condition.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    dit_mod, vae_mod, fm = import_reference()
    gen_flow_match(fm)
    dit_case(dit_mod, "tiny_t2v", synth.TINY_DIT, (3, 4, 6), 20, 13, 637.5, 100)
    dit_case(dit_mod, "small_t2v", synth.SMALL_DIT, (2, 5, 7), 24, 24, 991.7355, 150)
    dit_case(dit_mod, "tiny_i2v", synth.TINY_DIT_I2V, (2, 4, 4), 16, 10, 92.5926, 200)
    gen_denoise(dit_mod, fm)
    gen_vae(vae_mod)
    gen_image_condition(vae_mod)
    gen_teacache(dit_mod, fm)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
