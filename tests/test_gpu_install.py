"""-m gpu: the drop-in boundary.  `svi_hip.install(pipe)` is exercised against a stand-in for the reference's
pipeline module: a module object holding a `model_fn_wan_video` global, a pipeline class defined in it, and a sampler
that — like pipelines/svi_video.py:401-408 — looks `model_fn_wan_video` up through the module globals at call time and
passes `self.dit` (an nn.Module whose state_dict() has the reference's keys) plus keyword inputs.  The reference itself
cannot be imported on the GPU box, so the stand-in reproduces its *interface* (attribute names install() reads, state-dict
keys, call form), not its arithmetic; the arithmetic is checked against the committed golden vectors of the real reference.
"""
import sys
import types

import numpy as np
import pytest
import torch
from torch import nn

import synth
from gpu_util import dev, errs, report
from test_oracle_dit import CASES, inputs

pytestmark = pytest.mark.gpu


class _Node(nn.Module):
    """Container that indexes like nn.ModuleList / nn.Sequential (`blocks[0]`, `text_embedding[0]`, len(blocks))."""

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


def module_from_state_dict(sd, device, dtype):
    root = _Node()
    for k, v in sd.items():
        parts, m = k.split("."), root
        for p in parts[:-1]:
            if p not in m._modules:
                m.add_module(p, _Node())
            m = m._modules[p]
        m.register_parameter(parts[-1], nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).to(device=device, dtype=dtype), requires_grad=False))
    return root


def wan_model_double(c, seed):
    """nn.Module with the reference WanModel's state-dict keys and the attributes WanDiT.from_module reads
    (models/wan_video_dit.py:407-470: dim, freq_dim, has_image_input, patch_embedding Conv3d, blocks[i].{ffn_dim,num_heads,norm1.eps})."""
    sd = synth.dit_state_dict(seed, **c)
    m = module_from_state_dict(sd, "cuda", torch.bfloat16)
    m.dim, m.freq_dim, m.has_image_input = c["dim"], c["freq_dim"], c["has_image_input"]
    m.patch_embedding.in_channels, m.patch_embedding.kernel_size = c["in_dim"], tuple(c["patch_size"])
    m.text_embedding[0].in_features = c["text_dim"]
    m.head.head.out_features = sd["head.head.weight"].shape[0]
    for i in range(len(m.blocks)):
        b = m.blocks[i]
        b.ffn_dim, b.num_heads, b.norm1 = c["ffn_dim"], synth.num_heads_of(c), types.SimpleNamespace(eps=1e-6)
    return m, sd


SAMPLER_SRC = '''
def model_fn_wan_video(dit, x, timestep, context, clip_feature=None, y=None, **kwargs):
    raise AssertionError("the PyTorch model_fn was called: install() did not take effect")


class SVIVideoPipeline:
    def __init__(self, dit, vae):
        self.dit, self.vae = dit, vae

    def one_cfg_step(self, latents, timestep, prompt_emb_posi, prompt_emb_nega, image_emb, cfg_scale):
        # call form of the reference sampler: module-global lookup, dit passed positionally, embeddings as keyword dicts
        posi = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_posi, **image_emb)
        nega = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_nega, **image_emb)
        return nega + cfg_scale * (posi - nega), posi, nega
'''


@pytest.fixture()
def pipeline_module():
    name = "svi_video_double"
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    exec(compile(SAMPLER_SRC, name + ".py", "exec"), mod.__dict__)
    mod.SVIVideoPipeline.__module__ = name
    yield mod
    del sys.modules[name]


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
def test_install_swaps_model_fn_and_matches_golden(golden, pipeline_module, name):
    import svi_hip
    c, grid, nt, nv, ts, seed = CASES[name]
    g = golden(f"dit_{name}.npz")
    dit, sd = wan_model_double(c, seed)
    pipe = pipeline_module.SVIVideoPipeline(dit, None)
    with pytest.raises(AssertionError):                      # before install(): the module's own function is what runs
        pipeline_module.model_fn_wan_video(dit, None, None, None)
    svi_hip.install(pipe, vae=False)
    assert pipeline_module.model_fn_wan_video is not pipeline_module._svi_hip_original_model_fn
    assert sorted(dit.state_dict()) == sorted(sd)            # pipe.dit is untouched: loaders / LoRA merge still see every key

    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    lat, t = dev(x), torch.tensor([ts], device="cuda")
    ctx_p = dev(ctx)
    ctx_n = dev(-np.asarray(ctx))                            # any other prompt (a permutation of the tokens would not do: attention is permutation-invariant)
    img = {k: dev(v) for k, v in kw.items()}
    guided, posi, nega = pipe.one_cfg_step(lat, t, {"context": ctx_p}, {"context": ctx_n}, img, 5.0)
    r16, _, _ = errs(posi, g["out_bf16"])
    r32, _, _ = errs(posi, g["out_fp32"])
    report("install_model_fn", case=name, vs_ref_bf16=r16, vs_ref_fp32=r32)
    assert posi.shape == g["out_fp32"].shape and r16 < 2e-2
    assert not torch.equal(posi, nega) and torch.isfinite(guided.float()).all()
    # same call again (next step of the clip, same prompt tensors): served with the cached context projections, identical bits
    _, posi2, nega2 = pipe.one_cfg_step(lat, t, {"context": ctx_p}, {"context": ctx_n}, img, 5.0)
    assert torch.equal(posi, posi2) and torch.equal(nega, nega2)
    # an in-place edit of the prompt embedding must not be served from the cache
    ctx_p.mul_(0.5)
    _, posi3, _ = pipe.one_cfg_step(lat, t, {"context": ctx_p}, {"context": ctx_n}, img, 5.0)
    assert not torch.equal(posi, posi3)
    direct = svi_hip.WanDiT.from_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    assert torch.equal(posi3, direct.forward(lat, t, ctx_p, **img))


def test_install_refuses_cpu_or_fp32_dit(pipeline_module):
    import svi_hip
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    dit, _ = wan_model_double(c, seed)
    with pytest.raises(RuntimeError):
        svi_hip.install(pipeline_module.SVIVideoPipeline(dit.float(), None), vae=False)
    other, _ = wan_model_double(c, seed)
    with pytest.raises(RuntimeError):                        # a WanModel that never went through install(): no silent PyTorch fallback
        svi_hip.pipeline._hip_model_fn(other, None, None, torch.zeros(1, device="cuda"))


def test_install_rebinds_vae_methods(golden, pipeline_module):
    import svi_hip
    g = golden("vae.npz")
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    dit, _ = wan_model_double(c, seed)
    vsd = synth.vae_state_dict(500)                          # the weights the golden vectors were generated with (tests/gen_golden.py)
    vae = module_from_state_dict(vsd, "cuda", torch.float32)
    pipe = pipeline_module.SVIVideoPipeline(dit, vae)
    svi_hip.install(pipe)
    z = torch.from_numpy(synth.randn(501, 1, 16, 3, 4, 6))[0].cuda()
    out = pipe.vae.decode([z], device="cuda")[0]             # signature of models/wan_video_vae.py:777
    r, mx, _ = errs(out, g["decode_3f"])
    assert r < 2e-5 and mx < 2e-4, (r, mx)
    vid = torch.from_numpy(np.tanh(synth.randn(503, 3, 9, 32, 48))).cuda()
    lat = pipe.vae.encode([vid], device="cuda")[0]           # models/wan_video_vae.py:759
    r, mx, _ = errs(lat, g["encode_9f"])
    assert r < 2e-5 and mx < 2e-4, (r, mx)


def test_install_routes_the_encoders(golden, pipeline_module):
    """install(pipe) with a text encoder (bf16, GPU), a prompter and an image encoder on the pipe: the prompter's call
    `self.text_encoder(ids, mask)` (prompters/wan_prompter.py:109) and `pipe.image_encoder.encode_image([frame])` (svi_video.py:317)
    land on the HIP encoders; the results are the reference's (golden t5_encoder.npz / clip_encoder.npz)."""
    import svi_hip
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    dit, _ = wan_model_double(c, seed)
    pipe = pipeline_module.SVIVideoPipeline(dit, None)
    pipe.text_encoder = module_from_state_dict(synth.t5_state_dict(synth.T5_SEED, **synth.T5_TINY), "cuda", torch.bfloat16)
    name, L, valid, cseed = synth.T5_TINY_CASES[0]
    ids, mask = synth.t5_ids(cseed, L, valid, synth.T5_TINY["vocab"])

    class Prompter:                          # call form of WanPrompter.encode_prompt (:99-112)
        text_encoder = pipe.text_encoder

        def encode_prompt(self, prompt, positive=True, device="cuda"):
            i, m = torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device)
            seq_lens = m.gt(0).sum(dim=1).long()
            prompt_emb = self.text_encoder(i, m)
            for k, v in enumerate(seq_lens):
                prompt_emb[:, v:] = 0
            return prompt_emb

    pipe.prompter = Prompter()
    vis = module_from_state_dict(synth.clip_state_dict(synth.CLIP_SEED, **synth.CLIP_TINY), "cuda", torch.bfloat16)
    vis.num_heads, vis.norm_eps = synth.CLIP_TINY["num_heads"], 1e-5
    ie = _Node()
    ie.add_module("model", _Node())
    ie.model.add_module("visual", vis)
    ie.encode_image = lambda videos: (_ for _ in ()).throw(AssertionError("the PyTorch image encoder was called"))
    pipe.image_encoder = ie
    svi_hip.install(pipe, vae=False)
    emb = pipe.prompter.encode_prompt("a prompt")
    g = golden("t5_encoder.npz")
    want = g[f"{name}_fp32"].copy()
    want[valid:] = 0
    assert emb.dtype == torch.bfloat16 and errs(emb[0], want)[0] < 1.2e-2 and not bool(emb[0, valid:].any())
    # the image encoder's parameters live in bf16 on the pipe; SVI casts the module to fp32 around the call: fp32 arithmetic on the
    # bf16-rounded parameters
    cname, shape, iseed = synth.CLIP_TINY_CASES[0]
    img = torch.from_numpy(synth.clip_image(iseed, *shape)).cuda()
    got = pipe.image_encoder.encode_image([img])
    sdr = {k: torch.from_numpy(v).to(torch.bfloat16).float() for k, v in synth.clip_state_dict(synth.CLIP_SEED, **synth.CLIP_TINY).items()}
    direct = svi_hip.WanImageEncoder.from_state_dict(sdr, num_heads=2).encode_image([img])
    assert got.dtype == torch.float32 and torch.equal(got, direct)
    from oracle import encoders_oracle as eo
    with torch.no_grad():
        want_img = eo.clip_encode_image(sdr, img.cpu(), synth.CLIP_TINY)
    assert errs(got, want_img)[0] < 2e-5                                    # the oracle (pinned to the reference) on the same bf16-rounded parameters
    assert errs(got, golden("clip_encoder.npz")[cname])[0] < 1e-1          # and the fp32-parameter golden within what that rounding moves (4.4e-2)


def test_checkpoint_and_lora_file_on_the_device(tmp_path):
    """SURVEY §8f N4 end to end: safetensors shards -> HBM -> bound WanDiT (svi_hip.checkpoint.load_dit), then a PEFT-style LoRA file merged
    on the device (svi_hip.lora.load_lora_, name matching = the reference's get_name_dict): the forward equals the forward of a model
    whose weights were patched with the reference's arithmetic  W + alpha * (B @ A)  in bf16 on the host."""
    from safetensors.torch import save_file
    import svi_hip
    from svi_hip import checkpoint, lora
    c, seed = synth.TINY_DIT, 100
    sd = {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in synth.dit_state_dict(seed, **c).items()}
    keys = list(sd)
    save_file({k: sd[k] for k in keys[: len(keys) // 2]}, str(tmp_path / "m1.safetensors"))
    save_file({k: sd[k] for k in keys[len(keys) // 2:]}, str(tmp_path / "m2.safetensors"))
    cfg = dict(eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    m = checkpoint.load_dit([str(tmp_path / "m1.safetensors"), str(tmp_path / "m2.safetensors")], cfg)
    ref = svi_hip.WanDiT.from_state_dict(sd, **cfg)
    x = torch.from_numpy(synth.randn(seed + 1, 1, 16, 3, 8, 12)).cuda()
    ctx = torch.from_numpy(synth.text_context(seed + 2, 20, c["text_dim"], 13)).cuda()
    t = torch.tensor([637.5])
    assert torch.equal(m.forward(x, t, ctx), ref.forward(x, t, ctx))
    auto = checkpoint.load_dit([str(tmp_path / "m1.safetensors"), str(tmp_path / "m2.safetensors")])       # constructor arguments read off the shapes
    assert checkpoint.infer_dit_config(sd) == dict(cfg, patch_size=tuple(c["patch_size"]), has_image_input=False)
    assert torch.equal(auto.forward(x, t, ctx), ref.forward(x, t, ctx))
    r = 8
    targets = ["blocks.0.self_attn.q.weight", "blocks.1.ffn.0.weight"]
    lsd, patched = {}, dict(sd)
    for i, name in enumerate(targets):
        o, inn = sd[name].shape
        up = torch.from_numpy(0.05 * synth.randn(900 + i, o, r)).to(torch.bfloat16)
        down = torch.from_numpy(0.05 * synth.randn(910 + i, r, inn)).to(torch.bfloat16)
        base = "diffusion_model." + name[: -len(".weight")]
        lsd[base + ".lora_B.default.weight"], lsd[base + ".lora_A.default.weight"] = up, down
        patched[name] = sd[name] + 0.7 * torch.mm(up, down)               # models/lora.py:259-262 in the model dtype
    assert lora.load_lora_(m, {k: v.cuda() for k, v in lsd.items()}, alpha=0.7) == 2
    want = svi_hip.WanDiT.from_state_dict(patched, **cfg).forward(x, t, ctx)
    got = m.forward(x, t, ctx)
    rel = float((got.float() - want.float()).norm() / want.float().norm())
    assert rel < 2e-3, rel            # the merge differs from torch.mm only in fp32 summation order on bf16 ties
    assert not torch.equal(got, ref.forward(x, t, ctx))


def test_load_vae_reads_the_stock_pth_checkpoint(tmp_path, golden):
    """Wan2.1_VAE.pth as the reference ships it: a torch pickle of the bare VideoVAE_ state dict (no "model." prefix), optionally wrapped
    in {"model_state": ...} (WanVideoVAEStateDictConverter.from_civitai, wan_video_vae.py:802-808).  Both forms, and the same weights as
    safetensors, load through svi_hip.checkpoint.load_vae and decode to the reference's golden output."""
    from safetensors.torch import save_file
    from svi_hip import checkpoint
    g = golden("vae.npz")
    vsd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    bare = {k[len("model."):]: v for k, v in vsd.items()}
    torch.save(bare, str(tmp_path / "Wan2.1_VAE.pth"))
    torch.save({"model_state": bare}, str(tmp_path / "wrapped.pth"))
    save_file({k: v.contiguous() for k, v in vsd.items()}, str(tmp_path / "vae.safetensors"))
    import svi_hip
    ref = svi_hip.WanVideoVAE.from_state_dict(vsd)
    z = torch.from_numpy(synth.randn(41, 16, 2, 8, 8)).cuda()
    want = ref.decode([z], device="cuda")[0]
    for name in ("Wan2.1_VAE.pth", "wrapped.pth", "vae.safetensors"):
        v = checkpoint.load_vae(str(tmp_path / name))
        assert torch.equal(v.decode([z], device="cuda")[0], want), name


REAL_SAMPLER_SRC = '''
def model_fn_wan_video(dit, x, timestep, context, clip_feature=None, y=None, **kwargs):
    raise AssertionError("the PyTorch model_fn was called: install() did not take effect")


class SVIVideoPipeline:
    def __init__(self, dit, scheduler):
        self.dit, self.vae, self.scheduler, self.device = dit, None, scheduler, "cuda"

    # signature and body follow pipelines/svi_video.py:392-421 (the call statements are the reference's; tests/test_reference_keys.py
    # runs install()'s routing on the reference's OWN source, compiled with ast, on the build box)
    def _sample_with_regular_video(self, latents, prompt_emb_posi, prompt_emb_nega, image_emb, extra_input, tea_cache_posi, tea_cache_nega, usp_kwargs, use_controlnet, cfg_scale, progress_bar_cmd):
        for progress_id, timestep in enumerate(progress_bar_cmd(self.scheduler.timesteps)):
            timestep = timestep.unsqueeze(0).to(device=self.device)
            if cfg_scale['text'] != 1.0:
                noise_pred_cond = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_posi, **image_emb, **extra_input, **tea_cache_posi, **usp_kwargs, use_controlnet=use_controlnet)
                noise_pred_uncond = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_nega, **image_emb, **extra_input, **tea_cache_nega, **usp_kwargs, use_controlnet=use_controlnet)
                noise_pred = noise_pred_uncond + cfg_scale['text'] * (noise_pred_cond - noise_pred_uncond)
            else:
                noise_pred = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_posi, **image_emb, **extra_input, **tea_cache_posi, **usp_kwargs, use_controlnet=use_controlnet)
            latents = self.scheduler.step(noise_pred, self.scheduler.timesteps[progress_id], latents)
        return latents
'''


@pytest.mark.parametrize("name,scale", [("tiny_t2v", 5.0), ("tiny_i2v", 5.0), ("tiny_t2v", 1.0)])
def test_install_sampler_is_the_denoise_loop_and_keeps_the_reference_bits(name, scale):
    """install(sampler=True) rebinds `_sample_with_regular_video` to the DenoiseLoop-backed sampler (both forwards of a step in one C
    call, fused CFG + Euler kernel — what bench.py times).  It must give the bits of the reference-style loop (two swapped
    model_fn_wan_video calls, torch's bf16 CFG arithmetic, the scheduler's tensor update) that install(sampler=False) leaves in place."""
    import svi_hip
    c, grid, nt, nv, ts, seed = CASES[name]
    modname = "svi_video_double2"
    mod = types.ModuleType(modname)
    sys.modules[modname] = mod
    try:
        exec(compile(REAL_SAMPLER_SRC, modname + ".py", "exec"), mod.__dict__)
        mod.SVIVideoPipeline.__module__ = modname
        dit, sd = wan_model_double(c, seed)
        sch = svi_hip.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        sch.set_timesteps(4, shift=5.0)
        x, ctx, kw = inputs(c, grid, nt, nv, seed)
        lat = dev(x)
        posi, nega = {"context": dev(ctx)}, {"context": dev(-np.asarray(ctx))}
        img = {k: dev(v) for k, v in kw.items()}
        args = (posi, nega, img, {}, {"tea_cache": None}, {"tea_cache": None}, {}, False, {"text": scale}, lambda it: it)
        slow_pipe = mod.SVIVideoPipeline(dit, sch)
        svi_hip.install(slow_pipe, vae=False, sampler=False)
        before = lat.clone()
        slow = slow_pipe._sample_with_regular_video(lat, *args)
        fast_pipe = mod.SVIVideoPipeline(dit, sch)
        svi_hip.install(fast_pipe, vae=False)
        assert fast_pipe._sample_with_regular_video.__func__ is svi_hip.pipeline._hip_sample_with_regular_video
        fast = fast_pipe._sample_with_regular_video(lat, *args)
        assert torch.equal(lat, before)                      # the caller's latents are left alone, as by the reference's loop
        assert fast.dtype == slow.dtype == torch.bfloat16 and torch.isfinite(fast.float()).all()
        assert torch.equal(fast, slow)
        # TeaCache objects are honoured by the fast sampler too (svi_video.py:500-501): same skip pattern, same bits as the slow path
        tp, tn = svi_hip.TeaCache(4, 0.2, "Wan2.1-T2V-1.3B"), svi_hip.TeaCache(4, 0.2, "Wan2.1-T2V-1.3B")
        sp, sn = svi_hip.TeaCache(4, 0.2, "Wan2.1-T2V-1.3B"), svi_hip.TeaCache(4, 0.2, "Wan2.1-T2V-1.3B")
        a_t = (posi, nega, img, {}, {"tea_cache": tp}, {"tea_cache": tn}, {}, False, {"text": scale}, lambda it: it)
        a_s = (posi, nega, img, {}, {"tea_cache": sp}, {"tea_cache": sn}, {}, False, {"text": scale}, lambda it: it)
        assert torch.equal(fast_pipe._sample_with_regular_video(lat, *a_t), slow_pipe._sample_with_regular_video(lat, *a_s))
        # the rolling window's next clips (test_svi.py:424-476 re-enters __call__ per clip): new latents / prompt tensors each time, the reference loop's bits
        # each time — and the installed sampler's resident loop replays the step graph it captured once (DenoiseLoop.adopt)
        loop = fast_pipe._svi_hip_loop
        assert loop.resident
        caps = loop.captures
        wants = []
        clips = [(dev(np.asarray(x) * (0.5 + 0.25 * k)), {"context": dev(np.asarray(ctx) * (0.7 + 0.1 * k))}) for k in range(3)]
        for lat_k, posi_k in clips:
            wants.append(slow_pipe._sample_with_regular_video(lat_k, posi_k, *args[1:]))
        fast_pipe._sample_with_regular_video(clips[0][0], clips[0][1], *args[1:])          # (the slow calls above went through the same handle: re-establish the graph)
        caps = loop.captures
        for (lat_k, posi_k), want in zip(clips, wants):
            assert torch.equal(fast_pipe._sample_with_regular_video(lat_k, posi_k, *args[1:]), want)
        assert loop.captures == caps                         # three clips, no capture
    finally:
        del sys.modules[modname]


def test_install_refuses_offload_behind_its_back_and_rebinds_moved_parameters(pipeline_module):
    """install() borrows the DiT's parameters by pointer.  A parameter that is re-created on the device after install() (a loader, a dtype round trip)
    is re-bound at the next clip; one that left the device — `pipe.dit.cpu()`, the offload machinery — is refused with a RuntimeError, never computed on
    a stale copy.  enable_vram_management / enable_cpu_offload on the installed pipeline are no-ops."""
    import svi_hip
    from svi_hip import pipeline
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    dit, sd = wan_model_double(c, seed)
    pipe = pipeline_module.SVIVideoPipeline(dit, None)
    pipe.cpu_offload = True                                  # as after the reference's enable_vram_management (svi_video.py:241)
    pipe.enable_cpu_offload = types.MethodType(lambda self: setattr(self, "cpu_offload", True), pipe)
    pipe.enable_vram_management = types.MethodType(lambda self, num_persistent_param_in_dit=None: self.enable_cpu_offload(), pipe)
    svi_hip.install(pipe, vae=False)
    assert pipe.cpu_offload is False
    pipe.enable_vram_management(num_persistent_param_in_dit=6 * 10 ** 9)       # test_svi.py:351 after install(): nothing happens
    pipe.enable_cpu_offload()
    assert pipe.cpu_offload is False
    hip = pipe._svi_hip_dit
    pipeline._assert_resident(pipe, hip, full=True)
    x, ctx, _ = inputs(c, grid, nt, nv, seed)
    lat, t, cd = dev(x), torch.tensor([ts], device="cuda"), dev(ctx)
    before = pipeline_module.model_fn_wan_video(dit, lat, timestep=t, context=cd).clone()
    # a parameter re-created on the device: picked up (the new values are what the next clip computes with)
    w = dit.blocks[0].ffn._modules["0"].weight
    dit.blocks[0].ffn._modules["0"].weight = torch.nn.Parameter((w.data * 0.5).clone(), requires_grad=False)
    pipeline._assert_resident(pipe, hip, full=True)
    after = pipeline_module.model_fn_wan_video(dit, lat, timestep=t, context=cd)
    direct = svi_hip.WanDiT.from_state_dict(dict(dit.state_dict()), eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    assert not torch.equal(after, before) and torch.equal(after, direct.forward(lat, t, cd))
    # a parameter that left the device: refused
    dit.blocks[0].ffn._modules["0"].weight = torch.nn.Parameter(w.data.cpu(), requires_grad=False)
    with pytest.raises(RuntimeError, match="offloaded after install"):
        pipeline._assert_resident(pipe, hip, full=True)
    dit.blocks[0].ffn._modules["0"].weight = torch.nn.Parameter(w.data.clone(), requires_grad=False)
    pipe.cpu_offload = True                                  # the flag alone (the class's own enable_cpu_offload reached around the rebind)
    with pytest.raises(RuntimeError, match="install"):
        pipeline._assert_resident(pipe, hip)


# WanVideoPipeline (pipelines/wan_video.py): the step loop is inline in __call__.  The double keeps the reference's signature (:197-219), the order of its
# statements and — verbatim, checked against the reference's own source with ast on the build box (tests/test_reference_keys.py) — its `for` loop (:266-278),
# including the timestep cast to the pipeline's dtype; the model-side methods around it are reduced to what the loop needs.
WAN_CALL_SRC = '''
import torch


def model_fn_wan_video(dit, x, timestep, context, clip_feature=None, y=None, **kwargs):
    raise AssertionError("the PyTorch model_fn was called: install() did not take effect")


class TeaCache:
    def __init__(self, num_inference_steps, rel_l1_thresh, model_id):
        raise AssertionError("not used by this test")


class WanVideoPipeline:
    def __init__(self, dit, scheduler, prompts):
        self.dit, self.vae, self.scheduler, self.device, self.torch_dtype, self.prompts = dit, None, scheduler, "cuda", torch.bfloat16, prompts
        self.image_encoder = None
        self.decoded = []

    def check_resize_height_width(self, height, width):
        return height, width

    def generate_noise(self, shape, seed=None, device="cpu", dtype=torch.float16):
        generator = None if seed is None else torch.Generator(device).manual_seed(seed)
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    def load_models_to_device(self, names):
        pass

    def encode_prompt(self, prompt, positive=True):
        return {"context": self.prompts[prompt]}

    def prepare_extra_input(self, latents=None):
        return {}

    def decode_video(self, latents, tiled=True, tile_size=(34, 34), tile_stride=(18, 16)):
        self.decoded.append(latents)
        return [latents]

    def tensor2video(self, frames):
        return frames

    @torch.no_grad()
    def __call__(self, prompt, negative_prompt="", input_image=None, input_video=None, denoising_strength=1.0, seed=None, rand_device="cpu", height=480, width=832,
                 num_frames=81, cfg_scale=5.0, num_inference_steps=50, sigma_shift=5.0, tiled=True, tile_size=(30, 52), tile_stride=(15, 26),
                 tea_cache_l1_thresh=None, tea_cache_model_id="", progress_bar_cmd=lambda x: x, progress_bar_st=None):
        height, width = self.check_resize_height_width(height, width)
        tiler_kwargs = {"tiled": tiled, "tile_size": tile_size, "tile_stride": tile_stride}
        self.scheduler.set_timesteps(num_inference_steps, denoising_strength=denoising_strength, shift=sigma_shift)
        noise = self.generate_noise((1, 16, (num_frames - 1) // 4 + 1, height//8, width//8), seed=seed, device=rand_device, dtype=torch.float32)
        noise = noise.to(dtype=self.torch_dtype, device=self.device)
        latents = noise
        self.load_models_to_device(["text_encoder"])
        prompt_emb_posi = self.encode_prompt(prompt, positive=True)
        if cfg_scale != 1.0:
            prompt_emb_nega = self.encode_prompt(negative_prompt, positive=False)
        image_emb = {}
        extra_input = self.prepare_extra_input(latents)
        tea_cache_posi = {"tea_cache": TeaCache(num_inference_steps, rel_l1_thresh=tea_cache_l1_thresh, model_id=tea_cache_model_id) if tea_cache_l1_thresh is not None else None}
        tea_cache_nega = {"tea_cache": TeaCache(num_inference_steps, rel_l1_thresh=tea_cache_l1_thresh, model_id=tea_cache_model_id) if tea_cache_l1_thresh is not None else None}
        self.load_models_to_device(["dit"])
        for progress_id, timestep in enumerate(progress_bar_cmd(self.scheduler.timesteps)):
            timestep = timestep.unsqueeze(0).to(dtype=self.torch_dtype, device=self.device)

            # Inference
            noise_pred_posi = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_posi, **image_emb, **extra_input, **tea_cache_posi)
            if cfg_scale != 1.0:
                noise_pred_nega = model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_nega, **image_emb, **extra_input, **tea_cache_nega)
                noise_pred = noise_pred_nega + cfg_scale * (noise_pred_posi - noise_pred_nega)
            else:
                noise_pred = noise_pred_posi

            # Scheduler
            latents = self.scheduler.step(noise_pred, self.scheduler.timesteps[progress_id], latents)
        self.load_models_to_device(['vae'])
        frames = self.decode_video(latents, **tiler_kwargs)
        self.load_models_to_device([])
        frames = self.tensor2video(frames[0])
        return frames
'''


@pytest.mark.parametrize("scale", [5.0, 1.0])
def test_wan_video_pipeline_call_runs_the_fused_loop_and_keeps_the_reference_bits(scale):
    """VERDICT r5 missing 3 / next 5: WanVideoPipeline's step loop is inline in __call__ (wan_video.py:266-278), so install() gives the instance a one-method
    subclass whose __call__ steers the reference's own __call__ (svi_hip.pipeline._hip_wan_pipeline_call): every statement outside the loop is the
    reference's, the loop's steps run on DenoiseLoop.  Bits equal to the reference's own loop over the swapped model_fn_wan_video (install(sampler=False)) —
    including the bf16-ROUNDED timestep its :267 feeds the model — the caller's progress bar is driven, the shadows are gone after the call, and a second
    clip replays the step graph captured for the first."""
    import svi_hip
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    f, h, w = grid
    modname = "wan_video_double"
    mod = types.ModuleType(modname)
    sys.modules[modname] = mod
    try:
        exec(compile(WAN_CALL_SRC, modname + ".py", "exec"), mod.__dict__)
        mod.WanVideoPipeline.__module__ = modname
        dit, sd = wan_model_double(c, seed)
        _, ctx, _ = inputs(c, grid, nt, nv, seed)
        prompts = {"a": dev(ctx), "b": dev(np.asarray(ctx) * 0.7), "neg": dev(-np.asarray(ctx))}
        kw = dict(negative_prompt="neg", height=16 * h, width=16 * w, num_frames=4 * (f - 1) + 1, cfg_scale=scale, num_inference_steps=5)

        def pipeline(**inst):
            sch = svi_hip.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
            p = mod.WanVideoPipeline(dit, sch, prompts)
            svi_hip.install(p, vae=False, encoders=False, resident=False, **inst)
            return p
        slow = pipeline(sampler=False)
        assert type(slow) is mod.WanVideoPipeline
        want = [slow("a", seed=3, **kw), slow("b", seed=4, **kw)]
        # the bf16 timestep is a different number: the test is only meaningful if rounding it changes the schedule's values
        assert not torch.equal(slow.scheduler.timesteps.to(torch.bfloat16).float(), slow.scheduler.timesteps.float())
        fast = pipeline()
        assert type(fast) is not mod.WanVideoPipeline and isinstance(fast, mod.WanVideoPipeline) and type(fast).__call__ is svi_hip.pipeline._hip_wan_pipeline_call
        assert type(fast).__module__ == modname and mod.WanVideoPipeline.__call__ is not svi_hip.pipeline._hip_wan_pipeline_call      # the class itself is untouched
        seen = []

        def bar(it):
            for v in it:
                seen.append(float(v))
                yield v
        got = fast("a", seed=3, progress_bar_cmd=bar, **kw)
        assert torch.equal(got, want[0]) and got.dtype == torch.bfloat16
        assert seen == [float(t) for t in fast.scheduler.timesteps]                    # the caller's progress bar walked the clip's steps
        assert len(fast.decoded) == 1 and fast.decoded[0] is got                       # decode_video saw the denoised latents
        assert not any(n in fast.__dict__ for n in ("encode_prompt", "prepare_extra_input", "decode_video", "encode_image"))
        loop = fast._svi_hip_loop
        caps = loop.captures
        assert loop.resident and caps >= 1
        assert torch.equal(fast("b", seed=4, **kw), want[1])
        assert loop.captures == caps                                                   # the next clip replays the first clip's step graph
        # what the fast loop does not cover passes through to the reference's own loop (float32 latents: the pipeline's dtype switched)
        fast.torch_dtype = slow.torch_dtype = torch.float32
        assert torch.equal(fast("a", seed=3, **kw).float(), slow("a", seed=3, **kw).float())
    finally:
        del sys.modules[modname]
