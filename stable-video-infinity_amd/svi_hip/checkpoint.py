"""Checkpoint I/O for the hot path: safetensors shards straight into HBM, bound to the HIP handles.

The reference loads every tensor to the CPU with `safe_open` and converts it there (models/utils.py:72-79 load_state_dict_from_safetensors,
model_manager.py:653-687), then moves modules to the GPU.  With 288 GB of HBM nothing has to pass through a host-side model: the shards are
read directly to the device (the safetensors library does the file mapping — I/O plumbing, no arithmetic), cast once, and bound by
state-dict key.  What stays with the reference's loader: model-type detection by key hash, the T5 / CLIP encoders, tokenisers.
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


def load_dit(paths: Union[str, Sequence[str]], cfg: dict, device="cuda"):
    """A WanDiT bound to the tensors of a (sharded) safetensors checkpoint.  bf16 and float8_e4m3fn tensors are bound as stored (fp8
    through the exact bind-time cast); anything else is cast to bf16, the dtype the pipelines run the DiT in."""
    from .dit import WanDiT
    sd = load_safetensors(paths, device=device)
    sd = {k: (v if v.dtype in (torch.bfloat16, torch.float8_e4m3fn) else v.to(torch.bfloat16)).contiguous() for k, v in sd.items()}
    m = WanDiT(**cfg)
    m.bind(sd)
    return m


def load_vae(path: str, device="cuda"):
    """A HIP WanVideoVAE from Wan2.1_VAE.pth-style weights saved as safetensors; keys without the "model." prefix get it
    (WanVideoVAEStateDictConverter.from_civitai, models/wan_video_vae.py:799-808)."""
    from .vae import WanVideoVAE
    sd = load_safetensors(path, device=device, torch_dtype=torch.float32)
    if "model_state" in sd:
        sd = sd["model_state"]
    if not any(k.startswith("model.") for k in sd):
        sd = {"model." + k: v for k, v in sd.items()}
    return WanVideoVAE.from_state_dict(sd)
