"""The drop-in boundary against the REAL reference classes (row b): the C side's weight table (csrc/svi_dit.hip svi_dit_create, csrc/svi_vae.hip
declare_architecture) must accept exactly `WanModel(**cfg).state_dict()` / `WanVideoVAE().state_dict()` of the reference — every key, every shape,
nothing missing — for the constructor tables the reference ships (models/wan_video_dit.py:655-714).  The reference modules are built on the
meta device (no memory, no arithmetic) and bound with dummy 16-byte-aligned pointers: bind / check_bound never dereference.

Runs where /root/reference exists (the build container); skipped on the GPU box.  No GPU needed: create / bind / check_bound are host code.
"""
import ctypes as C
import os

import pytest
import torch

import synth

REF = os.environ.get("SVI_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "diffsynth")), reason="the reference checkout is not on this box")


@pytest.fixture(scope="module")
def ref():
    import gen_golden
    return gen_golden.import_reference()


# WanModelStateDictConverter.from_civitai config tables (wan_video_dit.py:655-714): 1.3B T2V, 14B T2V, 14B I2V
CONFIGS = {
    "1.3B-T2V": dict(has_image_input=False, patch_size=(1, 2, 2), in_dim=16, dim=1536, ffn_dim=8960, freq_dim=256, text_dim=4096, out_dim=16,
                     num_heads=12, num_layers=30, eps=1e-6),
    "14B-T2V": dict(has_image_input=False, patch_size=(1, 2, 2), in_dim=16, dim=5120, ffn_dim=13824, freq_dim=256, text_dim=4096, out_dim=16,
                    num_heads=40, num_layers=40, eps=1e-6),
    "14B-I2V": dict(has_image_input=True, patch_size=(1, 2, 2), in_dim=36, dim=5120, ffn_dim=13824, freq_dim=256, text_dim=4096, out_dim=16,
                    num_heads=40, num_layers=40, eps=1e-6),
    # the talk variant: the I2V table with the audio modules switched on (WanModel(enable_multitalk=True), wan_video_dit.py:421,455-470)
    "14B-I2V-talk": dict(has_image_input=True, patch_size=(1, 2, 2), in_dim=36, dim=5120, ffn_dim=13824, freq_dim=256, text_dim=4096, out_dim=16,
                         num_heads=40, num_layers=40, eps=1e-6, enable_multitalk=True),
}


def _bind_all(lib_fn, handle, state_dict, dtype_code):
    from svi_hip import _lib as L
    for i, (name, t) in enumerate(state_dict.items()):
        shape = (C.c_int64 * t.dim())(*t.shape)
        L.check(lib_fn(handle, name.encode(), C.c_void_p(0x10000 + 16 * i), dtype_code, shape, t.dim()), f"bind {name}")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_dit_weight_table_accepts_the_reference_state_dict(ref, name):
    from svi_hip import _lib as L
    dit_mod, _, _ = ref
    cfg = CONFIGS[name]
    with torch.device("meta"):
        m = dit_mod.WanModel(**cfg)
    sd = m.state_dict()
    c = L.DitConfig(cfg["dim"], cfg["in_dim"], cfg["ffn_dim"], cfg["out_dim"], cfg["text_dim"], cfg["freq_dim"], cfg["eps"], *cfg["patch_size"],
                    cfg["num_heads"], cfg["num_layers"], int(cfg["has_image_input"]), int(cfg.get("enable_multitalk", False)))
    h = C.c_void_p()
    L.check(L.lib().svi_dit_create(C.byref(c), C.byref(h)), "svi_dit_create")
    try:
        assert L.lib().svi_dit_check_bound(h) == 2                    # SVI_ERR_UNBOUND before anything is bound
        _bind_all(L.lib().svi_dit_bind_weight, h, sd, L.SVI_BF16)     # every reference key is known, every shape matches
        L.check(L.lib().svi_dit_check_bound(h), "svi_dit_check_bound")    # ... and nothing the C side needs is missing
    finally:
        L.lib().svi_dit_destroy(h)
    # the synthetic parameter inventory the parity tests are built on is the same table
    want = synth.dit_param_shapes(**{k: v for k, v in cfg.items() if k not in ("num_heads", "eps")})
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(want)


def test_dit_weight_table_rejects_a_wrong_shape_and_an_unknown_key(ref):
    from svi_hip import _lib as L
    cfg = CONFIGS["1.3B-T2V"]
    c = L.DitConfig(cfg["dim"], cfg["in_dim"], cfg["ffn_dim"], cfg["out_dim"], cfg["text_dim"], cfg["freq_dim"], cfg["eps"], *cfg["patch_size"],
                    cfg["num_heads"], cfg["num_layers"], 0)
    h = C.c_void_p()
    L.check(L.lib().svi_dit_create(C.byref(c), C.byref(h)), "svi_dit_create")
    try:
        bad = (C.c_int64 * 2)(1536, 1537)
        assert L.lib().svi_dit_bind_weight(h, b"blocks.0.self_attn.q.weight", C.c_void_p(0x10000), L.SVI_BF16, bad, 2) == 1
        ok = (C.c_int64 * 2)(1536, 1536)
        assert L.lib().svi_dit_bind_weight(h, b"blocks.0.cross_attn.k_img.weight", C.c_void_p(0x10000), L.SVI_BF16, ok, 2) == 1   # T2V has no image branch
        assert L.lib().svi_dit_bind_weight(h, b"blocks.30.self_attn.q.weight", C.c_void_p(0x10000), L.SVI_BF16, ok, 2) == 1
    finally:
        L.lib().svi_dit_destroy(h)


def test_vae_parameter_inventory_is_the_reference_state_dict(ref):
    """svi_vae_bind_weight launches packing kernels, so the VAE table is compared through the shared inventory instead: the key -> shape map
    the HIP side declares (svi_hip.vae.vae_param_shapes, mirrored by declare_architecture in csrc/svi_vae.hip and checked against it on the
    GPU by every WanVideoVAE.from_state_dict) equals the reference module's state dict."""
    from svi_hip.vae import vae_param_shapes
    _, vae_mod, _ = ref
    with torch.device("meta"):
        v = vae_mod.WanVideoVAE()
    sd = {k: tuple(t.shape) for k, t in v.state_dict().items()}
    assert sd == dict(vae_param_shapes())
    assert sd == dict(synth.vae_param_shapes())


def test_text_encoder_weight_table_accepts_the_reference_state_dict(ref):
    """WanTextEncoder() with its default constructor arguments = umT5-XXL as the Wan pipelines load it (wan_video_text_encoder.py:211-220):
    every key of its state dict binds (bf16), nothing the C side needs is missing, and the synthetic inventory of the parity tests is the
    same table."""
    import importlib
    from svi_hip import _lib as L
    te = importlib.import_module("diffsynth.models.wan_video_text_encoder")
    with torch.device("meta"):
        m = te.WanTextEncoder()
    sd = m.state_dict()
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32, shared_pos=False)
    assert (m.dim, m.dim_attn, m.dim_ffn, m.num_heads, m.num_layers, m.num_buckets, m.shared_pos) == tuple(cfg[k] for k in
                                                                                                         ("dim", "dim_attn", "dim_ffn", "num_heads", "num_layers", "num_buckets", "shared_pos"))
    c = L.T5Config(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"], cfg["num_buckets"], 128, 0)
    h = C.c_void_p()
    L.check(L.lib().svi_t5_create(C.byref(c), C.byref(h)), "svi_t5_create")
    try:
        assert L.lib().svi_t5_check_bound(h) == 2
        _bind_all(L.lib().svi_t5_bind_weight, h, sd, L.SVI_BF16)
        L.check(L.lib().svi_t5_check_bound(h), "svi_t5_check_bound")
        bad = (C.c_int64 * 2)(4096, 4097)
        assert L.lib().svi_t5_bind_weight(h, b"blocks.0.attn.q.weight", C.c_void_p(0x10000), L.SVI_BF16, bad, 2) == 1
        ok = (C.c_int64 * 2)(32, 64)
        assert L.lib().svi_t5_bind_weight(h, b"pos_embedding.embedding.weight", C.c_void_p(0x10000), L.SVI_BF16, ok, 2) == 1    # shared_pos is off
        assert L.lib().svi_t5_bind_weight(h, b"blocks.0.attn.q.weight", C.c_void_p(0x10000), L.SVI_F32, (C.c_int64 * 2)(4096, 4096), 2) == 1
    finally:
        L.lib().svi_t5_destroy(h)
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(synth.t5_param_shapes(**cfg))
    # T5RelativeEmbedding.max_dist default = the C side's
    assert m.blocks[0].pos_embedding.max_dist == 128


def test_image_encoder_weight_table_accepts_the_reference_state_dict(ref):
    """The visual tower exactly as clip_xlm_roberta_vit_h_14 configures it (wan_video_image_encoder.py:822-849, XLMRobertaCLIP :686-701) — the
    module WanImageEncoder holds as self.model.visual: all of its keys bind (fp32; post_norm / head / the last block are accepted and unused),
    and check_bound passes."""
    import importlib
    from svi_hip import _lib as L
    ie = importlib.import_module("diffsynth.models.wan_video_image_encoder")
    with torch.device("meta"):
        vis = ie.VisionTransformer(image_size=224, patch_size=14, dim=1280, mlp_ratio=4, out_dim=1024, num_heads=16, num_layers=32, pool_type="token",
                                   pre_norm=True, post_norm=False, activation="gelu", attn_dropout=0.0, proj_dropout=0.0, embedding_dropout=0.0, norm_eps=1e-5)
    sd = vis.state_dict()
    c = L.ClipConfig(224, 14, 1280, 4, 16, 32, 31, 1e-5)
    h = C.c_void_p()
    L.check(L.lib().svi_clip_create(C.byref(c), C.byref(h)), "svi_clip_create")
    try:
        assert L.lib().svi_clip_check_bound(h) == 2
        _bind_all(L.lib().svi_clip_bind_weight, h, sd, L.SVI_F32)
        L.check(L.lib().svi_clip_check_bound(h), "svi_clip_check_bound")
        tok, dim = C.c_int32(), C.c_int32()
        L.check(L.lib().svi_clip_tokens(h, C.byref(tok), C.byref(dim)), "svi_clip_tokens")
        assert (tok.value, dim.value) == (257, 1280)
        assert L.lib().svi_clip_bind_weight(h, b"transformer.32.norm1.weight", C.c_void_p(0x10000), L.SVI_F32, (C.c_int64 * 1)(1280), 1) == 1
        assert L.lib().svi_clip_bind_weight(h, b"patch_embedding.bias", C.c_void_p(0x10000), L.SVI_F32, (C.c_int64 * 1)(1280), 1) == 1   # pre_norm: no bias
    finally:
        L.lib().svi_clip_destroy(h)
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(synth.clip_param_shapes(image_size=224, patch_size=14, dim=1280, mlp_ratio=4, num_heads=16, num_layers=32))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_read_off_the_shapes_equals_the_reference_table(ref, name):
    """The reference looks WanModel's constructor arguments up by an md5 of the key names (wan_video_dit.py:655-714); svi_hip.checkpoint
    reads them off the tensor shapes.  For every table entry: build the reference model from the entry, infer from its state dict, get
    the entry back."""
    from svi_hip.checkpoint import infer_dit_config
    dit_mod, _, _ = ref
    cfg = CONFIGS[name]
    with torch.device("meta"):
        sd = dit_mod.WanModel(**cfg).state_dict()
    got = infer_dit_config(sd)
    want = dict(cfg)
    want.setdefault("enable_multitalk", False)
    got.setdefault("enable_multitalk", False)
    assert got == want


def test_config_tables_of_the_reference_converter(ref):
    """...and the tables themselves: every `config = {...}` literal of WanModelStateDictConverter.from_civitai / from_diffusers that
    describes a Wan2.1 model is reproduced (compiled out of the reference source, not retyped)."""
    import ast
    import inspect
    from svi_hip.checkpoint import infer_dit_config
    dit_mod, _, _ = ref
    src = open(os.path.join(REF, "diffsynth/models/wan_video_dit.py")).read()
    tables = []
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Dict) and getattr(node.targets[0], "id", "") == "config" and node.value.keys:
            tables.append(ast.literal_eval(node.value))
    assert len(tables) >= 4
    tables = [t for t in tables if "has_image_input" in t]       # the diffusers-style entry lacks a required argument: WanModel(**it) raises in the reference too
    assert len(tables) >= 4
    for t in tables:
        accepted = set(inspect.signature(dit_mod.WanModel.__init__).parameters)          # the diffusers-style table carries keys the class ignores
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in t.items() if k in accepted}
        with torch.device("meta"):
            sd = dit_mod.WanModel(**kw).state_dict()
        got = infer_dit_config(sd)
        for k, v in kw.items():
            assert got.get(k, False) == v, (k, got.get(k), v)


# ------------------------------------------------------------------------------------------------------------------
# install() against the real classes: what WanDiT.from_module reads off a WanModel INSTANCE, and the swaps
# install() makes, exercised on the reference's own sampler source.
@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_of_module_reads_the_real_wan_model(ref, name):
    """svi_hip.dit.config_of_module on the real WanModel (meta device) for the converter's constructor tables: every attribute it
    reads exists on the real class, and the constructor arguments come back."""
    from svi_hip.dit import config_of_module
    dit_mod, _, _ = ref
    cfg = CONFIGS[name]
    with torch.device("meta"):
        m = dit_mod.WanModel(**cfg)
    got = config_of_module(m)
    want = dict(cfg, patch_size=tuple(cfg["patch_size"]))
    want.setdefault("enable_multitalk", False)
    assert got == want


def _compiled_sampler_module(name="svi_video_compiled"):
    """A module object holding the reference's OWN `_sample_with_regular_video` source (compiled out of pipelines/svi_video.py with ast,
    nothing retyped) inside a class named as in the reference, plus a module-level `model_fn_wan_video` that must never run."""
    import ast
    import sys
    import types
    src = open(os.path.join(REF, "diffsynth/pipelines/svi_video.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SVIVideoPipeline")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_sample_with_regular_video")
    call = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__call__")
    new_cls = ast.ClassDef(name="SVIVideoPipeline", bases=[], keywords=[], body=[fn], decorator_list=[])
    mod_ast = ast.Module(body=[new_cls], type_ignores=[])
    ast.fix_missing_locations(mod_ast)
    mod = types.ModuleType(name)
    sys.modules[name] = mod

    def never(*a, **k):
        raise AssertionError("the PyTorch model_fn_wan_video ran: the swap did not take effect")
    mod.model_fn_wan_video = never
    exec(compile(mod_ast, name + ".py", "exec"), mod.__dict__)
    mod.SVIVideoPipeline.__module__ = name
    return mod, fn, call


def test_hip_sampler_has_the_reference_signature_and_call_form(ref):
    """The sampler install() binds takes exactly the parameters of SVIVideoPipeline._sample_with_regular_video (svi_video.py:392), in
    order — and __call__ hands them over positionally in that order (svi_video.py:506-509)."""
    import ast
    import inspect
    from svi_hip.pipeline import _hip_sample_with_regular_video
    _, fn, call = _compiled_sampler_module("svi_video_sig")
    ref_params = [a.arg for a in fn.args.args]
    assert list(inspect.signature(_hip_sample_with_regular_video).parameters) == ref_params
    stmt = next(n for n in ast.walk(call) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "_sample_with_regular_video")
    assert [a.id for a in stmt.args] == ref_params[1:] and not stmt.keywords


def test_install_routing_through_the_real_sampler_source(ref):
    """_route_dit (the DiT half of install()) on a pipeline whose class and module are the reference's own sampler source: the real call
    statements (`model_fn_wan_video(self.dit, latents, timestep=timestep, **prompt_emb_posi, **image_emb, **extra_input, **tea_cache_posi,
    **usp_kwargs, use_controlnet=use_controlnet)`, svi_video.py:401-408) reach the HIP twin's forward with the right tensors, twice per
    step, and the reference's own CFG / scheduler arithmetic runs around them; the sampler rebind keeps the original reachable."""
    import gen_golden
    from svi_hip import pipeline
    from svi_hip.dit import PromptPins
    dit_mod, _, fm = ref
    mod, _, _ = _compiled_sampler_module("svi_video_route")
    calls = []

    class FakeHip:                       # records what reaches WanDiT.forward; no GPU on this box
        _ctx_cache_on = False
        dim, patch_size = 8, (1, 2, 2)

        def context_cache(self, on):
            self._ctx_cache_on = bool(on)

        def forward(self, x, timestep, context, clip_feature=None, y=None, add_condition=None, **kw):
            calls.append(dict(x=x, timestep=timestep, context=context, clip_feature=clip_feature, y=y, add_condition=add_condition))
            return x * 0.5 + context.float().mean().to(x.dtype)

    with torch.device("meta"):
        real_dit = dit_mod.WanModel(**CONFIGS["1.3B-T2V"])
    pipe = mod.SVIVideoPipeline.__new__(mod.SVIVideoPipeline)
    pipe.dit, pipe.device = real_dit, "cpu"
    pipe.scheduler = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    pipe.scheduler.set_timesteps(3, shift=5.0)
    hip = FakeHip()
    pipeline._route_dit(pipe, hip, sampler=False)
    assert mod.model_fn_wan_video is pipeline._hip_model_fn and pipeline._INSTALLED[id(real_dit)] is hip
    lat = torch.randn(1, 16, 2, 4, 4).to(torch.bfloat16)
    pos, neg = torch.randn(1, 6, 8).to(torch.bfloat16), torch.randn(1, 6, 8).to(torch.bfloat16)
    out = pipe._sample_with_regular_video(lat, {"context": pos}, {"context": neg}, {}, {}, {"tea_cache": None}, {"tea_cache": None}, {},
                                          False, {"text": 5.0}, lambda x: x)
    assert len(calls) == 6 and out.shape == lat.shape and out.dtype == torch.bfloat16
    assert all(c["context"] is (pos if i % 2 == 0 else neg) for i, c in enumerate(calls))           # cond first, then uncond, every step
    assert [float(c["timestep"]) for c in calls[::2]] == [float(t) for t in pipe.scheduler.timesteps]
    assert calls[0]["x"] is lat and calls[2]["x"] is not lat                                          # the loop re-creates latents; step 0 sees the caller's
    # the same arithmetic by hand: CFG combine and the scheduler's step around the recorded forwards
    want = lat
    for i, t in enumerate(pipe.scheduler.timesteps):
        c = want * 0.5 + pos.float().mean().to(want.dtype)
        u = want * 0.5 + neg.float().mean().to(want.dtype)
        want = pipe.scheduler.step(u + 5.0 * (c - u), pipe.scheduler.timesteps[i], want)
    assert torch.equal(out, want)
    # with the sampler rebind: the instance attribute is ours, the class's own stays reachable for what the fast loop does not cover
    pipeline._route_dit(pipe, hip, sampler=True)
    assert pipe._sample_with_regular_video.__func__ is pipeline._hip_sample_with_regular_video
    assert pipe._svi_hip_original_sampler.__func__ is mod.SVIVideoPipeline._sample_with_regular_video
    del calls[:]
    out2 = pipe._sample_with_regular_video(lat, {"context": pos}, {"context": neg}, {}, {}, {"tea_cache": None}, {"tea_cache": None}, {},
                                           False, {"text": 5.0}, lambda x: x)       # CPU latents: not the fast loop's case -> the original sampler
    assert torch.equal(out2, want) and len(calls) == 6
    # the scalar the fused Euler kernel is handed equals what the reference scheduler multiplies by
    for i, t in enumerate(pipe.scheduler.timesteps):
        nxt = pipe.scheduler.sigmas[i + 1] if i + 1 < len(pipe.scheduler.sigmas) else 0.0
        assert pipeline._step_delta_of(pipe.scheduler, t) == float(nxt - pipe.scheduler.sigmas[i])
    import sys
    for n in ("svi_video_route", "svi_video_sig"):
        sys.modules.pop(n, None)


def test_two_speaker_audio_routing_is_unreachable_in_the_reference():
    """Row N3 remainder (VERDICT r3 missing #4): `SingleStreamMutiAttention.forward` has a two-speaker branch (models/attention.py:417-483:
    per-speaker 1-D RoPE classes from a reference-attention map), but every call site in the reference passes the literal `human_num=1` with
    `x_ref_attn_map=None` (models/wan_video_dit.py:364-365, models/wan_video_dit_talk.py:374-375), `SelfAttention.forward` is never handed
    `ref_target_masks` by a DiTBlock, and the talk pipeline prepares ONE speaker's audio (`audio_prepare_single`, pipelines/svi_video_talk.py:413).
    The HIP path therefore serves human_num == 1 — what the reference executes.  This test reads the reference's sources and fails the day a call
    site starts passing anything else, i.e. the day the branch becomes reachable and has to be built."""
    import ast
    sites = 0
    for rel in ("diffsynth/models/wan_video_dit.py", "diffsynth/models/wan_video_dit_talk.py"):
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "audio_cross_attn":
                kw = {k.arg: k.value for k in node.keywords}
                assert isinstance(kw.get("human_num"), ast.Constant) and kw["human_num"].value == 1, (rel, node.lineno)
                assert isinstance(kw.get("x_ref_attn_map"), ast.Constant) and kw["x_ref_attn_map"].value is None, (rel, node.lineno)
                sites += 1
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "self_attn":
                assert not any(k.arg == "ref_target_masks" for k in node.keywords) and len(node.args) <= 2, (rel, node.lineno)
    assert sites >= 2
    # nothing in the pipelines names the parameter at all
    import glob
    for f in glob.glob(os.path.join(REF, "diffsynth", "pipelines", "*.py")) + glob.glob(os.path.join(REF, "*.py")):
        src = open(f).read()
        assert "human_num" not in src and "ref_target_masks" not in src, f


# ---------------------------------------------------------------------------------------------------------------- offload neutralised
def _compiled_offload_pipeline(ref, name="svi_video_offload"):
    """A pipeline class made of the reference's OWN sources — SVIVideoPipeline.enable_vram_management (svi_video.py:156-241) and BasePipeline's
    enable_cpu_offload (base.py:105-106), compiled with ast, nothing retyped — over the real wrapper classes of vram_management/layers.py."""
    import ast
    import importlib
    import sys
    import types
    dit_mod, vae_mod, _ = ref
    vm = importlib.import_module("diffsynth.vram_management")
    te = importlib.import_module("diffsynth.models.wan_video_text_encoder")
    tree = ast.parse(open(os.path.join(REF, "diffsynth/pipelines/svi_video.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SVIVideoPipeline")
    evm = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "enable_vram_management")
    base = ast.parse(open(os.path.join(REF, "diffsynth/pipelines/base.py")).read())
    bcls = next(n for n in base.body if isinstance(n, ast.ClassDef) and n.name == "BasePipeline")
    eco = next(n for n in bcls.body if isinstance(n, ast.FunctionDef) and n.name == "enable_cpu_offload")
    mod_ast = ast.Module(body=[ast.ClassDef(name="SVIVideoPipeline", bases=[], keywords=[], body=[evm, eco], decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod_ast)
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    mod.__dict__.update(torch=torch, enable_vram_management=vm.enable_vram_management, AutoWrappedModule=vm.AutoWrappedModule,
                        AutoWrappedLinear=vm.AutoWrappedLinear, T5RelativeEmbedding=te.T5RelativeEmbedding, T5LayerNorm=te.T5LayerNorm,
                        RMSNorm=dit_mod.RMSNorm, RMS_norm=vae_mod.RMS_norm, CausalConv3d=vae_mod.CausalConv3d, Upsample=vae_mod.Upsample)
    exec(compile(mod_ast, name + ".py", "exec"), mod.__dict__)
    mod.SVIVideoPipeline.__module__ = name
    return mod, vm


def _offload_pipe(ref, mod):
    dit_mod, vae_mod, _ = ref
    te = __import__("importlib").import_module("diffsynth.models.wan_video_text_encoder")
    c = synth.TINY_DIT
    pipe = mod.SVIVideoPipeline.__new__(mod.SVIVideoPipeline)
    pipe.device, pipe.torch_dtype, pipe.cpu_offload = "cpu", torch.bfloat16, False
    pipe.dit = dit_mod.WanModel(eps=1e-6, num_heads=synth.num_heads_of(c), **c).to(torch.bfloat16)
    pipe.text_encoder = te.WanTextEncoder(vocab=64, dim=32, dim_attn=32, dim_ffn=64, num_heads=2, num_layers=1, num_buckets=8).to(torch.bfloat16)
    pipe.vae = vae_mod.WanVideoVAE()
    pipe.image_encoder = None
    return pipe


def test_enable_vram_management_is_harmless_before_and_after_install(ref, capsys):
    """test_svi.py:316-351 ends in `pipe.enable_vram_management(num_persistent_param_in_dit=args.num_persistent_param_in_dit)` — called
    unconditionally.  The residency half of install() (`_make_resident` + `_neutralise_offload`) against the reference's real wrapper
    machinery, in both orders: AFTER line 351 the wrappers are undone (checkpoint keys, the very same Parameter objects, cpu_offload off);
    BEFORE it the call statement itself — compiled out of test_svi.py — does nothing and says so once.  And the class's own function invoked
    on the instance afterwards is refused at the next clip."""
    import ast
    import sys
    import types
    from svi_hip import pipeline
    mod, vm = _compiled_offload_pipeline(ref)
    try:
        # the call statement of test_svi.py:351, as written there
        tree = ast.parse(open(os.path.join(REF, "test_svi.py")).read())
        stmt = next(n for n in ast.walk(tree) if isinstance(n, ast.Expr) and isinstance(n.value, ast.Call) and isinstance(n.value.func, ast.Attribute)
                    and n.value.func.attr == "enable_vram_management")
        assert 340 <= stmt.lineno <= 360
        line_351 = compile(ast.fix_missing_locations(ast.Module(body=[stmt], type_ignores=[])), "test_svi.py", "exec")
        args = types.SimpleNamespace(num_persistent_param_in_dit=6 * 10 ** 9)

        # ---- install AFTER line 351
        pipe = _offload_pipe(ref, mod)
        keys = {n: list(m.state_dict()) for n, m in (("dit", pipe.dit), ("vae", pipe.vae), ("text_encoder", pipe.text_encoder))}
        params = {n: p for n, p in pipe.dit.named_parameters()}
        exec(line_351, {"pipe": pipe, "args": args})
        assert pipe.cpu_offload and pipe.dit.vram_management_enabled
        assert any(type(m).__name__ == "AutoWrappedLinear" for m in pipe.dit.modules()) and list(pipe.dit.state_dict()) != keys["dit"]
        with pytest.raises(RuntimeError, match="VRAM management"):
            pipeline._assert_resident(pipe, None)
        pipeline._make_resident(pipe, device="cpu")
        pipeline._neutralise_offload(pipe)
        assert not pipe.cpu_offload and not pipe.dit.vram_management_enabled
        for n, m in (("dit", pipe.dit), ("vae", pipe.vae), ("text_encoder", pipe.text_encoder)):
            assert list(m.state_dict()) == keys[n], n
            assert not any(type(x).__name__ in pipeline._WRAPPERS for x in m.modules())
        assert all(p is params[n] for n, p in pipe.dit.named_parameters())          # nothing was copied: the same Parameter objects
        pipeline._assert_resident(pipe, None)

        # ---- install BEFORE line 351
        pipe = _offload_pipe(ref, mod)
        keys = list(pipe.dit.state_dict())
        pipeline._make_resident(pipe, device="cpu")
        pipeline._neutralise_offload(pipe)
        capsys.readouterr()
        exec(line_351, {"pipe": pipe, "args": args})
        exec(line_351, {"pipe": pipe, "args": args})
        err = capsys.readouterr().err
        assert err.count("enable_vram_management() is a no-op") == 1                   # says so, once
        pipe.enable_cpu_offload()
        assert not pipe.cpu_offload and list(pipe.dit.state_dict()) == keys and not getattr(pipe.dit, "vram_management_enabled", False)
        pipeline._assert_resident(pipe, None)

        # ---- the class's own function on the instance, behind install()'s back: refused at the next clip
        type(pipe).enable_vram_management(pipe, num_persistent_param_in_dit=None)
        with pytest.raises(RuntimeError, match="install"):
            pipeline._assert_resident(pipe, None)
    finally:
        sys.modules.pop("svi_video_offload", None)


def test_example_launcher_has_the_reference_cli_surface():
    """examples/test_svi_hip.py takes every argument the reference's test_svi.py defines (names and defaults read out of its parse_args with ast), so a
    command line written for the reference runs the HIP launcher unchanged (+ --synthetic for boxes without weights)."""
    import ast
    import importlib.util
    tree = ast.parse(open(os.path.join(REF, "test_svi.py")).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_args")
    ref_args = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            name = node.args[0].value
            kw = {k.arg: k.value for k in node.keywords}
            default = ast.literal_eval(kw["default"]) if "default" in kw and not isinstance(kw["default"], ast.BinOp) else None
            ref_args[name.lstrip("-")] = default
    assert len(ref_args) >= 25 and "num_motion_frames" in ref_args and "seed_times" in ref_args
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("test_svi_hip_example", os.path.join(root, "examples", "test_svi_hip.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ours = vars(mod.parse_args([]))
    missing = sorted(set(ref_args) - set(ours))
    assert not missing, missing
    for name, default in ref_args.items():
        if default is not None and name != "num_persistent_param_in_dit":
            assert ours[name] == default, (name, ours[name], default)
    assert mod.COMMON_NEGATIVE_PROMPT in open(os.path.join(REF, "test_svi.py")).read()       # the negative prompt is the reference's string
    assert mod.calculate_dimensions(1920, 1080, 832) == (464, 832) and mod.calculate_dimensions(640, 480, 832) == (480, 640)


def _wan_call_ast():
    import ast
    tree = ast.parse(open(os.path.join(REF, "diffsynth/pipelines/wan_video.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "WanVideoPipeline")
    return next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__call__")


def test_wan_pipeline_double_has_the_reference_loop_and_signature():
    """tests/test_gpu_install.py's WanVideoPipeline double (the GPU box has no reference): its __call__ takes the reference's parameters, in order, with the
    reference's defaults (but for the progress bar), and its step loop is the reference's `for` statement node for node (wan_video.py:266-278)."""
    import ast
    import test_gpu_install as tgi
    ref_call = _wan_call_ast()
    dbl_cls = next(n for n in ast.parse(tgi.WAN_CALL_SRC).body if isinstance(n, ast.ClassDef) and n.name == "WanVideoPipeline")
    dbl_call = next(n for n in dbl_cls.body if isinstance(n, ast.FunctionDef) and n.name == "__call__")
    assert [a.arg for a in dbl_call.args.args] == [a.arg for a in ref_call.args.args]
    ref_defaults = {a.arg: ast.dump(d) for a, d in zip(ref_call.args.args[-len(ref_call.args.defaults):], ref_call.args.defaults)}
    dbl_defaults = {a.arg: ast.dump(d) for a, d in zip(dbl_call.args.args[-len(dbl_call.args.defaults):], dbl_call.args.defaults)}
    ref_defaults.pop("progress_bar_cmd"), dbl_defaults.pop("progress_bar_cmd")            # tqdm there, identity here
    assert dbl_defaults == ref_defaults
    loops = [[n for n in ast.walk(fn) if isinstance(n, ast.For)] for fn in (ref_call, dbl_call)]
    assert len(loops[0]) == len(loops[1]) == 1 and ast.dump(loops[0][0]) == ast.dump(loops[1][0])
    # every self.<method>(...) the reference's __call__ makes outside the branches the double leaves out exists on the double under the same name
    called = {n.func.attr for n in ast.walk(ref_call) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute)
              and isinstance(n.func.value, ast.Name) and n.func.value.id == "self"}
    have = {n.name for n in dbl_cls.body if isinstance(n, ast.FunctionDef)}
    assert called - have <= {"preprocess_images", "encode_video", "encode_image"}, called - have


def test_wan_pipeline_call_steering_on_the_reference_source(ref):
    """_hip_wan_pipeline_call around the reference's REAL WanVideoPipeline.__call__ (compiled from its source with ast): the values the loop needs are picked up
    where the reference's own statements produce them, the shadows are gone afterwards, and — CPU latents are not the fused loop's case — the reference's own
    loop runs over the swapped model_fn_wan_video and sees the timestep in the pipeline's dtype (bf16-rounded)."""
    import ast
    import sys
    import types
    from svi_hip import pipeline
    dit_mod, _, fm = ref
    call = _wan_call_ast()
    name = "wan_video_compiled"
    stubs = ast.parse('''
def check_resize_height_width(self, height, width):
    return height, width
def generate_noise(self, shape, seed=None, device="cpu", dtype=None):
    import torch
    return torch.randn(shape, generator=torch.Generator(device).manual_seed(seed), device=device, dtype=dtype)
def load_models_to_device(self, names):
    self.loaded.append(list(names))
def encode_prompt(self, prompt, positive=True):
    return {"context": self.prompts[prompt]}
def prepare_extra_input(self, latents=None):
    return {}
def decode_video(self, latents, tiled=True, tile_size=(34, 34), tile_stride=(18, 16)):
    return [latents]
def tensor2video(self, frames):
    return frames
''').body
    cls = ast.ClassDef(name="WanVideoPipeline", bases=[], keywords=[], body=stubs + [call], decorator_list=[])
    mod_ast = ast.Module(body=[cls], type_ignores=[])
    ast.fix_missing_locations(mod_ast)
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    try:
        mod.torch, mod.tqdm = torch, (lambda x: x)

        def never(*a, **k):
            raise AssertionError("the PyTorch model_fn_wan_video ran: the swap did not take effect")
        mod.model_fn_wan_video = never
        mod.TeaCache = None
        exec(compile(mod_ast, name + ".py", "exec"), mod.__dict__)
        mod.WanVideoPipeline.__module__ = name
        calls = []

        class FakeHip:
            _ctx_cache_on = False
            dim, patch_size, ffn_dim, num_heads = 8, (1, 2, 2), 16, 1

            def context_cache(self, on):
                self._ctx_cache_on = bool(on)

            def forward(self, x, timestep, context, clip_feature=None, y=None, add_condition=None, **kw):
                calls.append(dict(timestep=timestep, context=context))
                return x * 0.5 + context.float().mean().to(x.dtype)

        with torch.device("meta"):
            real_dit = dit_mod.WanModel(**CONFIGS["1.3B-T2V"])
        pipe = mod.WanVideoPipeline.__new__(mod.WanVideoPipeline)
        pipe.dit, pipe.device, pipe.torch_dtype, pipe.image_encoder, pipe.loaded = real_dit, "cpu", torch.bfloat16, None, []
        pipe.prompts = {"p": torch.randn(1, 6, 8).to(torch.bfloat16), "n": torch.randn(1, 6, 8).to(torch.bfloat16)}
        pipe.scheduler = fm.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        hip = FakeHip()
        pipeline._route_dit(pipe, hip, sampler=True)
        assert type(pipe) is not mod.WanVideoPipeline and type(pipe).__call__ is pipeline._hip_wan_pipeline_call and type(pipe).__mro__[1] is mod.WanVideoPipeline
        bar_saw = []
        out = pipe("p", negative_prompt="n", seed=7, height=32, width=32, num_frames=5, num_inference_steps=3, tiled=False,
                   progress_bar_cmd=lambda it: (bar_saw.append(it), it)[1])
        assert bar_saw and bar_saw[0] is pipe.scheduler.timesteps                         # pass-through: the reference's loop took the caller's bar
        assert len(calls) == 6 and all(c["context"] is pipe.prompts["p" if i % 2 == 0 else "n"] for i, c in enumerate(calls))
        assert all(c["timestep"].dtype == torch.bfloat16 for c in calls)                 # :267 — the pipeline's dtype
        assert [float(c["timestep"]) for c in calls[::2]] == [float(t.to(torch.bfloat16)) for t in pipe.scheduler.timesteps]
        assert not any(n in pipe.__dict__ for n in ("encode_prompt", "prepare_extra_input", "decode_video", "encode_image"))
        # the reference's arithmetic by hand around the recorded forwards
        want = torch.randn((1, 16, 2, 4, 4), generator=torch.Generator("cpu").manual_seed(7), dtype=torch.float32).to(torch.bfloat16)
        for i, t in enumerate(pipe.scheduler.timesteps):
            c = want * 0.5 + pipe.prompts["p"].float().mean().to(want.dtype)
            u = want * 0.5 + pipe.prompts["n"].float().mean().to(want.dtype)
            want = pipe.scheduler.step(u + 5.0 * (c - u), pipe.scheduler.timesteps[i], want)
        assert torch.equal(out, want)
        assert pipe.loaded[-1] == [] and ["dit"] in pipe.loaded                          # the reference's own statements around the loop ran
        # a second install() does not stack subclasses
        pipeline._route_dit(pipe, hip, sampler=True)
        assert type(pipe).__mro__[1] is mod.WanVideoPipeline
    finally:
        sys.modules.pop(name, None)
