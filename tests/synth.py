"""Deterministic synthetic weights/inputs for the Wan DiT and the Wan VAE.

There are no checkpoints and no network here, so every parity test runs on seeded random
weights of the reference architectures.  Weights come from numpy's legacy MT19937
`RandomState` (bit-stable across numpy versions), NOT from torch's default init, so that the
golden generator (run once, in the container that has /root/reference) and the tests (run
anywhere) build bit-identical tensors without storing them.

Parameter names and shapes follow the reference state dicts:
  DiT: diffsynth/models/wan_video_dit.py:408-470 (WanModel.__init__), :321-336 (DiTBlock)
  VAE: diffsynth/models/wan_video_vae.py:276-326 (Encoder3d), :379-430 (Decoder3d), :512-517
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np


# ----------------------------------------------------------------------------------- DiT
def dit_param_shapes(dim: int, in_dim: int, ffn_dim: int, out_dim: int, text_dim: int, freq_dim: int,
                     patch_size: Tuple[int, int, int], num_layers: int, has_image_input: bool,
                     enable_multitalk: bool = False) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    s["patch_embedding.weight"] = (dim, in_dim, *patch_size)
    s["patch_embedding.bias"] = (dim,)
    lin("text_embedding.0", dim, text_dim)
    lin("text_embedding.2", dim, dim)
    lin("time_embedding.0", dim, freq_dim)
    lin("time_embedding.2", dim, dim)
    lin("time_projection.1", dim * 6, dim)
    for i in range(num_layers):
        b = f"blocks.{i}."
        s[b + "modulation"] = (1, 6, dim)
        for att in ("self_attn", "cross_attn"):
            for nm in ("q", "k", "v", "o"):
                lin(b + att + "." + nm, dim, dim)
            s[b + att + ".norm_q.weight"] = (dim,)
            s[b + att + ".norm_k.weight"] = (dim,)
            if att == "cross_attn" and has_image_input:
                lin(b + att + ".k_img", dim, dim)
                lin(b + att + ".v_img", dim, dim)
                s[b + att + ".norm_k_img.weight"] = (dim,)
        s[b + "norm3.weight"] = (dim,)
        s[b + "norm3.bias"] = (dim,)
        lin(b + "ffn.0", ffn_dim, dim)
        lin(b + "ffn.2", dim, ffn_dim)
        if enable_multitalk:                                  # wan_video_dit.py:338-351
            lin(b + "audio_cross_attn.q_linear", dim, dim)
            lin(b + "audio_cross_attn.proj", dim, dim)
            lin(b + "audio_cross_attn.kv_linear", 2 * dim, 768)
            s[b + "norm_x.weight"] = (dim,)
            s[b + "norm_x.bias"] = (dim,)
    s["head.modulation"] = (1, 2, dim)
    lin("head.head", out_dim * int(np.prod(patch_size)), dim)
    if has_image_input:
        s["img_emb.proj.0.weight"] = (1280,)
        s["img_emb.proj.0.bias"] = (1280,)
        lin("img_emb.proj.1", 1280, 1280)
        lin("img_emb.proj.3", dim, 1280)
        s["img_emb.proj.4.weight"] = (dim,)
        s["img_emb.proj.4.bias"] = (dim,)
    if enable_multitalk:                                      # AudioProjModel, wan_video_dit.py:455-470
        lin("audio_proj.proj1", 512, 5 * 12 * 768)
        lin("audio_proj.proj1_vf", 512, 8 * 12 * 768)
        lin("audio_proj.proj2", 512, 512)
        lin("audio_proj.proj3", 32 * 768, 512)
        s["audio_proj.norm.weight"] = (768,)
        s["audio_proj.norm.bias"] = (768,)
    return s


def _fill(rs: np.random.RandomState, name: str, shape: tuple) -> np.ndarray:
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "modulation":
        return (rs.standard_normal(shape) / math.sqrt(shape[-1])).astype(np.float32)
    if leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
        return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)      # norm gains
    if leaf == "bias" and ("norm" in name or name.startswith("img_emb.proj.0") or name.startswith("img_emb.proj.4")):
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        return rs.uniform(-bound, bound, size=shape).astype(np.float32)
    if leaf == "bias":
        return rs.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    raise KeyError(name)


def dit_state_dict(seed: int, **cfg) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    return OrderedDict((k, _fill(rs, k, shp)) for k, shp in dit_param_shapes(**cfg).items())


TINY_DIT = dict(dim=128, in_dim=16, ffn_dim=256, out_dim=16, text_dim=64, freq_dim=256,
                patch_size=(1, 2, 2), num_layers=2, has_image_input=False)
TINY_DIT_I2V = dict(dim=128, in_dim=36, ffn_dim=256, out_dim=16, text_dim=64, freq_dim=256,
                    patch_size=(1, 2, 2), num_layers=2, has_image_input=True)
SMALL_DIT = dict(dim=256, in_dim=16, ffn_dim=768, out_dim=16, text_dim=128, freq_dim=256,
                 patch_size=(1, 2, 2), num_layers=2, has_image_input=False)
WAN_1_3B = dict(dim=1536, in_dim=16, ffn_dim=8960, out_dim=16, text_dim=4096, freq_dim=256,
                patch_size=(1, 2, 2), num_layers=30, has_image_input=False)


WAN_14B_I2V = dict(dim=5120, in_dim=36, ffn_dim=13824, out_dim=16, text_dim=4096, freq_dim=256,
                   patch_size=(1, 2, 2), num_layers=40, has_image_input=True)      # wan_video_dit.py:699-712


def num_heads_of(cfg: dict) -> int:
    return cfg["dim"] // 128          # the 3-D RoPE split needs head_dim == 128


# ----------------------------------------------------------------------------------- VAE
def vae_param_shapes() -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    base, z, mult = 96, 16, (1, 2, 4, 4)

    def conv3(name, o, i, k=(3, 3, 3)):
        s[name + ".weight"] = (o, i, *k)
        s[name + ".bias"] = (o,)

    def conv2(name, o, i, k=3):
        s[name + ".weight"] = (o, i, k, k)
        s[name + ".bias"] = (o,)

    def res(p, i, o):
        s[p + "residual.0.gamma"] = (i, 1, 1, 1)
        conv3(p + "residual.2", o, i)
        s[p + "residual.3.gamma"] = (o, 1, 1, 1)
        conv3(p + "residual.6", o, o)
        if i != o:
            conv3(p + "shortcut", o, i, (1, 1, 1))

    def attn(p, c):
        s[p + "norm.gamma"] = (c, 1, 1)
        conv2(p + "to_qkv", 3 * c, c, 1)
        conv2(p + "proj", c, c, 1)

    # encoder
    e = "model.encoder."
    dims = [base * u for u in (1,) + mult]
    conv3(e + "conv1", dims[0], 3)
    idx, tdown = 0, (False, True, True)
    for i, (di, do) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(2):
            res(f"{e}downsamples.{idx}.", di, do); idx += 1
            di = do
        if i != len(mult) - 1:
            conv2(f"{e}downsamples.{idx}.resample.1", do, do)
            if tdown[i]:
                conv3(f"{e}downsamples.{idx}.time_conv", do, do, (3, 1, 1))
            idx += 1
    top = dims[-1]
    res(e + "middle.0.", top, top); attn(e + "middle.1.", top); res(e + "middle.2.", top, top)
    s[e + "head.0.gamma"] = (top, 1, 1, 1)
    conv3(e + "head.2", 2 * z, top)
    conv3("model.conv1", 2 * z, 2 * z, (1, 1, 1))
    conv3("model.conv2", z, z, (1, 1, 1))
    # decoder
    d = "model.decoder."
    dims = [base * u for u in (mult[-1],) + mult[::-1]]
    conv3(d + "conv1", dims[0], z)
    res(d + "middle.0.", dims[0], dims[0]); attn(d + "middle.1.", dims[0]); res(d + "middle.2.", dims[0], dims[0])
    idx, tup = 0, tdown[::-1]
    for i, (di, do) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            di = di // 2
        for _ in range(3):
            res(f"{d}upsamples.{idx}.", di, do); idx += 1
            di = do
        if i != len(mult) - 1:
            conv2(f"{d}upsamples.{idx}.resample.1", do // 2, do)
            if tup[i]:
                conv3(f"{d}upsamples.{idx}.time_conv", do * 2, do, (3, 1, 1))
            idx += 1
    s[d + "head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv3(d + "head.2", 3, dims[-1])
    return s


def vae_state_dict(seed: int) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    return OrderedDict((k, _fill(rs, k, shp)) for k, shp in vae_param_shapes().items())


# ----------------------------------------------------------------------------------- inputs
def randn(seed: int, *shape) -> np.ndarray:
    return np.random.RandomState(seed).standard_normal(shape).astype(np.float32)


def text_context(seed: int, tokens: int, text_dim: int, valid: int) -> np.ndarray:
    """[1, tokens, text_dim] with the padded tail zeroed, like diffsynth/prompters/wan_prompter.py:107-108."""
    c = randn(seed, 1, tokens, text_dim)
    c[:, valid:] = 0
    return c


def condition_frames(seed: int, n: int, height: int, width: int) -> np.ndarray:
    """uint8 RGB frames [n, height, width, 3] (what the clip loop hands to encode_images_adaptive as PIL images)."""
    return np.random.RandomState(seed).randint(0, 256, size=(n, height, width, 3)).astype(np.uint8)


def frames_to_tensor(frames_u8: np.ndarray) -> np.ndarray:
    """BasePipeline.preprocess_image (pipelines/base.py:44-45) per frame: float32(x) * (2 / 255) - 1, [n, 3, H, W]."""
    return (frames_u8.astype(np.float32) * (2 / 255) - 1).transpose(0, 3, 1, 2)


# (name, number of condition frames, ref_pad_cfg, ref_pad_num): the cases of tests/golden/image_condition.npz
IMAGE_CONDITION_CASES = [("first_only_zero_pad", 1, False, 0), ("first_only_ref_everywhere", 1, False, -1),
                         ("two_motion_frames_ref_pad2_cfg", 2, True, 2), ("five_motion_frames_zero_pad", 5, False, 0)]


# tests/golden/teacache_tiny.npz (tests/gen_golden.py::gen_teacache): steps, threshold, model id and the timesteps used
TEA_STEPS, TEA_THRESH, TEA_MODEL = 8, 0.01, "Wan2.1-T2V-1.3B"
TEA_TIMESTEPS = [500.0, 500.01, 500.03, 500.035, 500.075, 500.08, 500.15, 500.155]


# ----------------------------------------------------------------------------------- fixtures at BASELINE sizes (gen_golden.py)
C1_SEED, C1_VIDEO_STRIDE = 900, 5            # golden/c1_e2e.npz: 1.3B weights seed; decoded video kept on a stride-5 lattice
C1_50_KEEP = (1, 5, 10, 20, 30, 40, 50)      # golden/c1_50step.npz: the C1 grid through FIFTY CFG-5 steps; latents kept after these steps
C2_VIDEO_STRIDE = 7                          # golden/vae_c2.npz
B14_SEED, B14_GRID = 950, (3, 20, 36)        # golden/dit_block_14b.npz: one 14B-I2V block on 2160 tokens


DEPTH_GRID, DEPTH_STEPS = (5, 30, 52), 2     # golden/dit_depth.npz: the 30-layer 1.3B model (C1_SEED weights) on 7800 tokens (>= 2048: the long-sequence
                                             # attention kernel and the 256^2 GEMM), one forward + a 2-step CFG loop
C2_GRID = (21, 30, 52)                       # BASELINE configs[1]: 81 frames at 832x480 -> latent [16,21,60,104] -> 32760 tokens
B13C2_SEED, B14C2_SEED = 970, 980            # golden/dit_block_c2.npz / dit_block_14b_c2.npz: ONE reference DiTBlock (1.3B / 14B-I2V widths) at L = 32760
C2_LOOP_STEPS = 2                            # golden/dit_c2_loop.npz: a 2-step CFG loop of the same model on the same latent
C2_FULL_STRIDE = 3                           # golden/dit_c2_full.npz: the reference's 30-layer 1.3B forward at C2 (C1_SEED weights), output kept on a stride-3 (h, w) lattice
C4_SEED, C4_LAYERS = 960, 4                  # golden/dit_c4_4blocks.npz: 4 of the 40 blocks of Wan2.1-I2V-14B end to end (in_dim-36 patchify, img_emb, head) on B14_GRID


def bf16_bits(a) -> np.ndarray:
    """fp32 array holding bf16-representable values -> their 16-bit patterns (fixtures store bf16 results at half the size)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return (a.view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b) -> np.ndarray:
    return (np.asarray(b, dtype=np.uint32) << 16).view(np.float32)


def B14_ROWS(L: int):
    return list(range(0, L, 67))


def C2_ROWS(L: int, stride: int = 244):
    """Token rows the C2-size block fixtures keep: a stride that is a multiple of 61 (coprime with the 52-wide row and the 1560-token frame
    of the C2 grid, so the rows walk through the (h, w) residues and every frame is hit) plus the last row of the ragged last 256-row
    tile.  1.3B widths: stride 244 (136 rows); 14B widths: stride 488 (69 rows) — the fixtures stay at 1-2 MB."""
    return list(range(0, L, stride)) + [L - 1]


# (name, latent shape, tile_size, tile_stride, seed) / (name, video shape, tile_size, tile_stride, seed): golden/vae_tiled.npz
TILED_DECODE_CASES = [("3x3", (16, 2, 8, 8), (4, 4), (2, 2), 520),
                      ("ragged", (16, 2, 9, 11), (4, 6), (3, 4), 521)]
TILED_ENCODE_CASES = [("3x4", (3, 5, 64, 80), (4, 4), (2, 2), 530)]


# golden/clip_stream.npz: the reference's clip loop + __call__ on a tiny I2V stream (gen_golden.py::gen_clip_stream)
STREAM_HW, STREAM_CLIP_SEED, STREAM_PROMPT_SEED, STREAM_IMAGE_SEED = (32, 48), 33, 40, 31
STREAM_CASES = [
    dict(name="m1", num_motion_frames=1, num_frames=17, num_clips=3, steps=2, num_prompts=3, prompt_repeat_times=1,
         use_first_prompt_only=False, ref_pad_cfg=False, ref_pad_num=-1),
    dict(name="m5", num_motion_frames=5, num_frames=17, num_clips=4, steps=2, num_prompts=2, prompt_repeat_times=2,
         use_first_prompt_only=False, ref_pad_cfg=True, ref_pad_num=2),
]


# ----------------------------------------------------------------------------------- dance pose embedder (golden/pose_embed.npz)
def pose_param_shapes(hidden: int = 16, dim: int = 5120) -> "OrderedDict[str, tuple]":
    """State dict of the nn.Sequential at pipelines/svi_video_dance.py:255-269 (convolutions at indices 0, 2, ..., 12)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    for i in range(7):
        cin, cout = (3 if i == 0 else hidden), (dim if i == 6 else hidden)
        k = (1, 2, 2) if i == 6 else (3, 3, 3)
        s[f"{2 * i}.weight"] = (cout, cin, *k)
        s[f"{2 * i}.bias"] = (cout,)
    return s


def pose_state_dict(seed: int, hidden: int = 16, dim: int = 5120) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    return OrderedDict((k, _fill(rs, k, shp)) for k, shp in pose_param_shapes(hidden, dim).items())


POSE_SEED = 700
POSE_CASES = [("f9", (3, 9, 32, 48), 701), ("f5_odd", (3, 5, 48, 80), 702)]        # (name, humanpose_data shape, seed)


def pose_video(seed: int, *shape) -> np.ndarray:
    """A pose video as the loader hands it over: float32 values 0..255 (sparse coloured strokes on black)."""
    rs = np.random.RandomState(seed)
    v = rs.randint(0, 256, size=shape).astype(np.float32)
    return v * (rs.uniform(size=shape) < 0.3)


# key styles of LoRA files (PEFT with / without an adapter name, with / without the "diffusion_model." prefix, an unrelated key)
LORA_KEY_EXAMPLES = [
    "diffusion_model.blocks.0.self_attn.q.lora_A.default.weight", "diffusion_model.blocks.0.self_attn.q.lora_B.default.weight",
    "blocks.3.ffn.0.lora_A.weight", "blocks.3.ffn.0.lora_B.weight",
    "diffusion_model.blocks.29.cross_attn.k_img.lora_A.weight", "diffusion_model.blocks.29.cross_attn.k_img.lora_B.weight",
    "blocks.1.self_attn.o.lora_B.adapter2.weight", "blocks.1.self_attn.o.lora_A.adapter2.weight",
    "dwpose_embedding.0.weight",
]


# ----------------------------------------------------------------------------------- talk variant (golden/dit_tiny_talk.npz, talk_sampler.npz)
TINY_DIT_TALK = dict(dim=256, in_dim=36, ffn_dim=512, out_dim=16, text_dim=64, freq_dim=256, patch_size=(1, 2, 2), num_layers=2,
                     has_image_input=True, enable_multitalk=True)
TALK_SEED, TALK_GRID = 800, (3, 4, 6)


def audio_windows(seed: int, frames: int):
    """audio_embed_tuple of one clip (svi_video_talk.py:425-444): first-frame window [1, 1, 5, 12, 768], later frames [1, f-1, 8, 12, 768]."""
    return 0.5 * randn(seed, 1, 1, 5, 12, 768), 0.5 * randn(seed + 1, 1, frames - 1, 8, 12, 768)


# ----------------------------------------------------------------------------------- prompt-side encoders (golden/t5_*.npz, clip_*.npz)
T5_TINY = dict(vocab=200, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32, shared_pos=False)
T5_XXL_BLOCK = dict(vocab=512, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=1, num_buckets=32, shared_pos=False)   # text_encoder:211-220 widths
T5_DEEP = dict(vocab=200, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=24, num_buckets=32, shared_pos=False)      # the real depth at the tiny width
T5_DEEP_CASE = ("deep", 48, 31, 907)
T5_SEED = 900
T5_TINY_CASES = [("short", 24, 13, 901), ("long", 160, 160, 902), ("one", 16, 1, 903), ("mid", 100, 77, 904)]      # (name, L, valid tokens, seed)
T5_XXL_CASE = ("xxl", 64, 40, 905)
T5_XXL_ROWS = [0, 1, 7, 19, 38, 39, 40, 63]


def t5_param_shapes(vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos) -> "OrderedDict[str, tuple]":
    """WanTextEncoder.state_dict() (wan_video_text_encoder.py:209-238)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["token_embedding.weight"] = (vocab, dim)
    if shared_pos:
        s["pos_embedding.embedding.weight"] = (num_buckets, num_heads)
    for i in range(num_layers):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (dim,)
        for n in "qkv":
            s[p + f"attn.{n}.weight"] = (dim_attn, dim)
        s[p + "attn.o.weight"] = (dim, dim_attn)
        s[p + "norm2.weight"] = (dim,)
        s[p + "ffn.gate.0.weight"] = (dim_ffn, dim)
        s[p + "ffn.fc1.weight"] = (dim_ffn, dim)
        s[p + "ffn.fc2.weight"] = (dim, dim_ffn)
        if not shared_pos:
            s[p + "pos_embedding.embedding.weight"] = (num_buckets, num_heads)
    s["norm.weight"] = (dim,)
    return s


def t5_state_dict(seed: int, **cfg) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for k, shp in t5_param_shapes(**cfg).items():
        if k.endswith("pos_embedding.embedding.weight"):
            out[k] = rs.standard_normal(shp).astype(np.float32)            # a bias of order 1 per bucket: a wrong bucket is visible
        elif k == "token_embedding.weight":
            out[k] = rs.standard_normal(shp).astype(np.float32)
        else:
            out[k] = _fill(rs, k, shp)
    return out


def t5_ids(seed: int, L: int, valid: int, vocab: int):
    """(ids, mask) as the tokenizer hands them over: `valid` tokens then pad id 0."""
    rs = np.random.RandomState(seed)
    ids = np.zeros((1, L), np.int64)
    ids[0, :valid] = rs.randint(1, vocab, size=valid)
    mask = np.zeros((1, L), np.int64)
    mask[0, :valid] = 1
    return ids, mask


CLIP_TINY = dict(image_size=28, patch_size=14, dim=160, mlp_ratio=4, num_heads=2, num_layers=3)
CLIP_H_BLOCK = dict(image_size=224, patch_size=14, dim=1280, mlp_ratio=4, num_heads=16, num_layers=2)     # image_encoder:826-834 widths, 2 of 32 blocks
CLIP_DEEP = dict(image_size=28, patch_size=14, dim=160, mlp_ratio=4, num_heads=2, num_layers=32)          # the real depth (31 blocks used) at the tiny width
CLIP_DEEP_CASE = ("deep", (1, 3, 36, 44), 957)
CLIP_SEED = 950
CLIP_TINY_CASES = [("down", (1, 3, 40, 56), 951), ("up", (1, 3, 20, 24), 952), ("batch2", (2, 3, 28, 28), 953)]      # (name, image shape, seed)
CLIP_H_CASE = ("h14", (1, 3, 480, 832), 955)
CLIP_H_ROWS = [0, 1, 2, 15, 16, 17, 100, 128, 200, 255, 256]


def clip_param_shapes(image_size, patch_size, dim, mlp_ratio, num_heads, num_layers) -> "OrderedDict[str, tuple]":
    """VisionTransformer.state_dict() as XLMRobertaCLIP builds it (wan_video_image_encoder.py:386-454: pool 'token', pre_norm, no
    patch-embedding bias)."""
    n = (image_size // patch_size) ** 2 + 1
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["cls_embedding"] = (1, 1, dim)
    s["pos_embedding"] = (1, n, dim)
    s["patch_embedding.weight"] = (dim, 3, patch_size, patch_size)
    s["pre_norm.weight"] = (dim,)
    s["pre_norm.bias"] = (dim,)
    for i in range(num_layers):
        p = f"transformer.{i}."
        s[p + "norm1.weight"] = (dim,); s[p + "norm1.bias"] = (dim,)
        s[p + "attn.to_qkv.weight"] = (3 * dim, dim); s[p + "attn.to_qkv.bias"] = (3 * dim,)
        s[p + "attn.proj.weight"] = (dim, dim); s[p + "attn.proj.bias"] = (dim,)
        s[p + "norm2.weight"] = (dim,); s[p + "norm2.bias"] = (dim,)
        s[p + "mlp.0.weight"] = (dim * mlp_ratio, dim); s[p + "mlp.0.bias"] = (dim * mlp_ratio,)
        s[p + "mlp.2.weight"] = (dim, dim * mlp_ratio); s[p + "mlp.2.bias"] = (dim,)
    s["post_norm.weight"] = (dim,)
    s["post_norm.bias"] = (dim,)
    s["head"] = (dim, 1024 if dim == 1280 else 64)
    return s


def clip_state_dict(seed: int, **cfg) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for k, shp in clip_param_shapes(**cfg).items():
        if k in ("cls_embedding", "pos_embedding", "head"):
            out[k] = (rs.standard_normal(shp) / math.sqrt(cfg["dim"])).astype(np.float32)
        elif k.endswith("to_qkv.weight"):
            out[k] = (rs.standard_normal(shp) * (3.0 / math.sqrt(cfg["dim"]))).astype(np.float32)      # sharp (non-uniform) attention rows
        else:
            out[k] = _fill(rs, k, shp)
    return out


def clip_image(seed: int, *shape) -> np.ndarray:
    """An image batch as preprocess_image hands it over: fp32 in [-1, 1], smooth + texture."""
    rs = np.random.RandomState(seed)
    b, c, h, w = shape
    yy, xx = np.meshgrid(np.linspace(0, 3, h), np.linspace(0, 4, w), indexing="ij")
    base = np.sin(yy[None, None] * (1 + np.arange(c)[None, :, None, None])) * np.cos(xx[None, None] + np.arange(b)[:, None, None, None])
    return np.clip(0.7 * base + 0.3 * rs.uniform(-1, 1, size=shape), -1, 1).astype(np.float32)
