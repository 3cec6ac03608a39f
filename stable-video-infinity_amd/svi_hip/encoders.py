"""Prompt-side encoders on libsvi_hip: they run once per clip and feed the DiT's `context` / `clip_feature`.

    WanTextEncoder     <- diffsynth/models/wan_video_text_encoder.py:209-256 (umT5-XXL encoder), called by
                          WanPrompter.encode_prompt (prompters/wan_prompter.py:99-112) as `self.text_encoder(ids, mask)`
    WanImageEncoder    <- diffsynth/models/wan_video_image_encoder.py:852-880, `encode_image(videos)`, called by
                          encode_images_adaptive (pipelines/svi_video.py:317) with the module switched to fp32 (:307-309)

The tokenizer (HuggingfaceTokenizer over the umt5-xxl sentencepiece files) stays the reference's: token ids are this boundary's input.
Dropout is the identity (the pipelines run the encoders in eval mode).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional

import torch

from . import _lib as L


def relative_position_buckets(num_buckets: int, max_dist: int, length: int) -> List[int]:
    """T5RelativeEmbedding._relative_position_bucket (text_encoder:175-194, bidirectional) for rel = key - query in
    -(length-1) .. length-1, as the C side tabulates it (host only: no GPU needed)."""
    out = (C.c_int32 * (2 * length - 1))()
    L.check(L.lib().svi_t5_relative_buckets(num_buckets, max_dist, length, out), "svi_t5_relative_buckets")
    return list(out)


def _bind_all(fn, handle, state_dict: Dict[str, torch.Tensor], dtype: torch.dtype, code: int, keep: Dict[str, torch.Tensor], what: str) -> None:
    for name, t in state_dict.items():
        if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
            raise RuntimeError(f"{what} parameter {name} must be a contiguous CUDA {dtype} tensor")
        shape = (C.c_int64 * t.dim())(*t.shape)
        L.check(fn(handle, name.encode(), t.data_ptr(), code, shape, t.dim()), f"bind {name}")
        keep[name] = t


class WanTextEncoder:
    """Mirror of the reference class of the same name; parameters are bf16 and borrowed."""

    def __init__(self, vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32, shared_pos=False,
                 max_dist=128):
        self.vocab, self.dim, self.dim_attn, self.dim_ffn = vocab, dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.shared_pos = num_heads, num_layers, num_buckets, shared_pos
        cfg = L.T5Config(vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, max_dist, int(bool(shared_pos)))
        h = C.c_void_p()
        L.check(L.lib().svi_t5_create(C.byref(cfg), C.byref(h)), "svi_t5_create")
        self._h = h
        self._params: Dict[str, torch.Tensor] = {}

    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor], device="cuda", **overrides) -> "WanTextEncoder":
        sd = state_dict
        layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        shared = "pos_embedding.embedding.weight" in sd
        pe = sd["pos_embedding.embedding.weight" if shared else "blocks.0.pos_embedding.embedding.weight"]
        cfg = dict(vocab=sd["token_embedding.weight"].shape[0], dim=sd["token_embedding.weight"].shape[1],
                   dim_attn=sd["blocks.0.attn.q.weight"].shape[0], dim_ffn=sd["blocks.0.ffn.fc1.weight"].shape[0],
                   num_heads=pe.shape[1], num_layers=layers, num_buckets=pe.shape[0], shared_pos=shared)
        cfg.update(overrides)
        m = cls(**cfg)
        m.bind({k: v.to(device=device, dtype=torch.bfloat16).contiguous() for k, v in sd.items()})
        return m

    @classmethod
    def from_module(cls, module) -> "WanTextEncoder":
        """An existing reference WanTextEncoder on the GPU in bf16: parameters are borrowed in place."""
        return cls.from_state_dict(dict(module.state_dict()))

    def bind(self, state_dict: Dict[str, torch.Tensor]) -> None:
        _bind_all(L.lib().svi_t5_bind_weight, self._h, state_dict, torch.bfloat16, L.SVI_BF16, self._params, "text encoder")
        L.check(L.lib().svi_t5_check_bound(self._h), "svi_t5_check_bound")

    def forward(self, ids: torch.Tensor, mask: Optional[torch.Tensor] = None, rows: str = "all") -> torch.Tensor:
        """ids [B, L] integer, mask [B, L] (tokenizer padding: a prefix of ones) -> [B, L, dim] bf16.
        rows="all": every row as text_encoder(ids, mask) returns it; rows="valid": only the unpadded rows are computed, the rest is zero —
        exactly what encode_prompt keeps (prompter:110-111) at a fraction of the work."""
        if ids.dim() != 2:
            raise ValueError(f"ids must be [B, L] (got {tuple(ids.shape)})")
        B, Ln = ids.shape
        ids = ids.to(device="cuda", dtype=torch.int64).contiguous()
        if bool(((ids < 0) | (ids >= self.vocab)).any()):
            raise IndexError("token id out of range of the embedding table")
        if mask is None:
            valid = [Ln] * B
        else:
            if tuple(mask.shape) != (B, Ln):
                raise ValueError(f"mask must be [B, L] = {(B, Ln)} (got {tuple(mask.shape)})")
            mk = mask.to("cpu").gt(0)
            valid = [int(v) for v in mk.sum(dim=1)]
            for b, v in enumerate(valid):
                if v < 1 or not bool(mk[b, :v].all()):
                    raise ValueError("mask must be a non-empty prefix mask (tokenizer padding on the right)")
        out = torch.empty((B, Ln, self.dim), dtype=torch.bfloat16, device=ids.device)
        for b in range(B):
            r = Ln if rows == "all" else valid[b]
            L.check(L.lib().svi_t5_forward(self._h, L.ptr(ids[b]), Ln, valid[b], r, L.ptr(out[b]), L.current_stream()), "svi_t5_forward")
        return out

    __call__ = forward

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().svi_t5_destroy(self._h)
                self._h = None
        except Exception:
            pass


class WanImageEncoder:
    """Mirror of the reference class of the same name (the visual tower of open-clip XLM-R ViT-H/14, all but the last block)."""

    def __init__(self, image_size=224, patch_size=14, dim=1280, mlp_ratio=4, num_heads=16, num_layers=32, layers_used=None, norm_eps=1e-5):
        self.image_size, self.patch_size, self.dim, self.num_heads, self.num_layers = image_size, patch_size, dim, num_heads, num_layers
        used = num_layers - 1 if layers_used is None else layers_used                      # use_31_block=True (image_encoder:877)
        cfg = L.ClipConfig(image_size, patch_size, dim, mlp_ratio, num_heads, num_layers, used, norm_eps)
        h = C.c_void_p()
        L.check(L.lib().svi_clip_create(C.byref(cfg), C.byref(h)), "svi_clip_create")
        self._h = h
        self._params: Dict[str, torch.Tensor] = {}
        self.tokens = (image_size // patch_size) ** 2 + 1

    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor], device="cuda", **cfg) -> "WanImageEncoder":
        """Keys of WanImageEncoder ("model.visual.…"), of the open-clip checkpoint ("visual.…", image_encoder:894-902) or of the bare
        VisionTransformer; anything outside the visual tower is ignored.  Parameters are converted to fp32 copies owned here — the
        reference casts the module to fp32 for the call and back afterwards (svi_video.py:307-309, :361-362)."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("model.visual."):
                k = k[len("model.visual."):]
            elif k.startswith("visual."):
                k = k[len("visual."):]
            elif k.startswith(("model.", "textual.", "log_scale")):
                continue
            sd[k] = v
        pw = sd["patch_embedding.weight"]
        layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("transformer."))
        auto = dict(patch_size=pw.shape[-1], dim=pw.shape[0], num_layers=layers,
                    mlp_ratio=sd["transformer.0.mlp.0.weight"].shape[0] // pw.shape[0],
                    image_size=int(round((sd["pos_embedding"].shape[1] - 1) ** 0.5)) * pw.shape[-1])
        auto.update(cfg)
        m = cls(**auto)
        m.bind({k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in sd.items()})
        return m

    @classmethod
    def from_module(cls, module, **cfg) -> "WanImageEncoder":
        vis = module.model.visual
        return cls.from_state_dict(dict(vis.state_dict()), num_heads=vis.num_heads, norm_eps=vis.norm_eps, **cfg)

    def bind(self, state_dict: Dict[str, torch.Tensor]) -> None:
        _bind_all(L.lib().svi_clip_bind_weight, self._h, state_dict, torch.float32, L.SVI_F32, self._params, "image encoder")
        L.check(L.lib().svi_clip_check_bound(self._h), "svi_clip_check_bound")

    def encode_image(self, videos: Iterable[torch.Tensor]) -> torch.Tensor:
        """videos: list of [b, 3, H, W] tensors in [-1, 1] -> fp32 [sum b, tokens, dim] (hidden states after the last-but-one block)."""
        outs = []
        for u in videos:
            if u.dim() != 4 or u.shape[1] != 3:
                raise ValueError(f"each image batch must be [b, 3, H, W] (got {tuple(u.shape)})")
            x = u.to(device="cuda", dtype=torch.float32).contiguous()
            b, _, H, W = x.shape
            out = torch.empty((b, self.tokens, self.dim), dtype=torch.float32, device=x.device)
            L.check(L.lib().svi_clip_encode_image(self._h, L.ptr(x), b, H, W, L.ptr(out), L.current_stream()), "svi_clip_encode_image")
            outs.append(out)
        return torch.cat(outs) if len(outs) != 1 else outs[0]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().svi_clip_destroy(self._h)
                self._h = None
        except Exception:
            pass
