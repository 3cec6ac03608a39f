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
