"""Checkpoint I/O for the hot path: safetensors shards straight into HBM, bound to the HIP handles.

The reference loads every tensor to the CPU with `safe_open` and converts it there (models/utils.py:72-79 load_state_dict_from_safetensors,
model_manager.py:653-687), then moves modules to the GPU.  With 288 GB of HBM nothing has to pass through a host-side model: the shards are
read directly to the device (the safetensors library does the file mapping — I/O plumbing, no arithmetic), cast once, and bound by
state-dict key.  The reference recognises a model by an md5 over its key names and looks the constructor arguments up in a table
(models/wan_video_dit.py:643-714, hash_state_dict_keys); `infer_dit_config` reads the same arguments off the tensor shapes instead, so a
checkpoint needs no table entry.  Tokenisers stay with the reference.
"""
from __future__ import annotations

from typing import Dict, Sequence, Union

import torch


def load_safetensors(paths: Union[str, Sequence[str]], device="cuda", torch_dtype=None) -> Dict[str, torch.Tensor]:
    """All tensors of one or more .safetensors shards on `device` (models/utils.py:72-79 semantics: later shards override earlier keys;
    `torch_dtype` casts every tensor, as load_state_dict(..., torch_dtype) does)."""
    from safetensors import safe_open
    if isinstance(paths, str):
        paths = [paths]
    out: Dict[str, torch.Tensor] = {}
    for p in paths:
        with safe_open(p, framework="pt", device=str(device)) as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[k] = t if torch_dtype is None else t.to(torch_dtype)
    return out


def infer_dit_config(shapes: Dict[str, Sequence[int]], patch_size=(1, 2, 2), eps: float = 1e-6) -> dict:
    """Constructor arguments of WanModel from a state dict's key -> shape map: what WanModelStateDictConverter.from_civitai returns for the
    checkpoints it knows by hash (wan_video_dit.py:655-714), for any checkpoint of the architecture.  head_dim is 128 (the 3-D RoPE split)."""
    shp = {k: tuple(getattr(v, "shape", v)) for k, v in shapes.items()}
    dim, in_dim = shp["patch_embedding.weight"][0], shp["patch_embedding.weight"][1]
    if tuple(shp["patch_embedding.weight"][2:]) != tuple(patch_size):
        patch_size = tuple(shp["patch_embedding.weight"][2:])
    layers = 1 + max(int(k.split(".")[1]) for k in shp if k.startswith("blocks."))
    pp = patch_size[0] * patch_size[1] * patch_size[2]
    cfg = dict(has_image_input="img_emb.proj.1.weight" in shp, patch_size=tuple(patch_size), in_dim=in_dim, dim=dim,
               ffn_dim=shp["blocks.0.ffn.0.weight"][0], freq_dim=shp["time_embedding.0.weight"][1], text_dim=shp["text_embedding.0.weight"][1],
               out_dim=shp["head.head.weight"][0] // pp, num_heads=dim // 128, num_layers=layers, eps=eps)
    if any(k.startswith("audio_proj.") for k in shp):
        cfg["enable_multitalk"] = True
    return cfg


def load_dit(paths: Union[str, Sequence[str]], cfg: dict = None, device="cuda"):
    """A WanDiT bound to the tensors of a (sharded) safetensors checkpoint.  bf16 and float8_e4m3fn tensors are bound as stored (fp8
    through the exact bind-time cast); anything else is cast to bf16, the dtype the pipelines run the DiT in.  cfg=None: read off the shapes."""
    from .dit import WanDiT
    sd = load_safetensors(paths, device=device)
    sd = {k: (v if v.dtype in (torch.bfloat16, torch.float8_e4m3fn) else v.to(torch.bfloat16)).contiguous() for k, v in sd.items()}
    if cfg is None:
        cfg = infer_dit_config(sd)
    m = WanDiT(**cfg)
    m.bind(sd)
    return m


def load_vae(path: str, device="cuda"):
    """A HIP WanVideoVAE from the stock Wan2.1_VAE.pth (a torch pickle, optionally wrapped in {"model_state": ...}) or the same weights
    as .safetensors; keys without the "model." prefix get it and everything is cast to fp32, the precision SVI runs the VAE in
    (WanVideoVAEStateDictConverter.from_civitai, models/wan_video_vae.py:797-808; svi_video.py:385-389)."""
    from .vae import WanVideoVAE
    sd = _load_any(path, device)
    if "model_state" in sd:
        sd = sd["model_state"]
    if not any(k.startswith("model.") for k in sd):
        sd = {"model." + k: v for k, v in sd.items()}
    sd = {k: v.to(device=device, dtype=torch.float32) for k, v in sd.items()}
    return WanVideoVAE.from_state_dict(sd)


def _load_any(path: str, device):
    """.safetensors shards or a torch checkpoint (the Wan encoders ship as .pth state dicts)."""
    if str(path).endswith(".safetensors"):
        return load_safetensors(path, device=device)
    return torch.load(path, map_location=device, weights_only=True)


def load_text_encoder(path: str, device="cuda"):
    """svi_hip.WanTextEncoder from models_t5_umt5-xxl-enc-bf16.pth-style weights (WanTextEncoder's own keys, bf16)."""
    from .encoders import WanTextEncoder
    return WanTextEncoder.from_state_dict(_load_any(path, device), device=device)


def load_image_encoder(path: str, device="cuda"):
    """svi_hip.WanImageEncoder from the open-clip XLM-R ViT-H/14 checkpoint ("visual." keys; the text tower is skipped) or from
    WanImageEncoder's own state dict ("model.visual." keys) — WanImageEncoderStateDictConverter.from_civitai, image_encoder:894-902."""
    from .encoders import WanImageEncoder
    return WanImageEncoder.from_state_dict(_load_any(path, device), device=device, num_heads=16)
