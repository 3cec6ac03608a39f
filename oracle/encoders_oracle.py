"""CPU oracle for the prompt-side encoders (umT5 text encoder, CLIP visual tower) — TEST INFRASTRUCTURE ONLY (see
oracle/wan_dit_oracle.py header for the import rule).

Parity status: PINNED.  tests/golden/t5_encoder.npz and clip_encoder.npz hold outputs of the reference's own WanTextEncoder (through
WanPrompter.encode_prompt, compiled out of prompters/wan_prompter.py) and of WanImageEncoder.encode_image (compiled out of
models/wan_video_image_encoder.py, around the reference's own VisionTransformer), produced by tests/gen_golden.py; the T5 relative
position bucket table is the output of T5RelativeEmbedding._relative_position_bucket itself.  tests/test_oracle_encoders.py checks
this restatement against them.

Restates:
  models/wan_video_text_encoder.py   :16-20 GELU, :24-35 T5LayerNorm, :39-84 T5Attention (no 1/sqrt(d) scale, bias added to the
                                     scores, softmax in fp32), :88-106 gated-GELU feed-forward, :110-139 block, :175-194 buckets,
                                     :239-249 forward;   prompters/wan_prompter.py:99-112 (rows >= the valid length are zeroed).
  models/wan_video_image_encoder.py  :864-880 encode_image (bicubic resize, *0.5+0.5, Normalize), :456-478 VisionTransformer.forward
                                     with use_31_block, :289-330 AttentionBlock (pre-norm), :234-268 SelfAttention.
`rounding="bf16"` restates where a bf16 module materialises bf16 tensors (every op output).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _r(rounding: Optional[str]):
    if rounding == "bf16":
        return lambda x: x.to(torch.bfloat16).float()
    return lambda x: x


# --------------------------------------------------------------------------------------------------------------------- T5
def relative_position_buckets(num_buckets: int, max_dist: int, length: int) -> np.ndarray:
    """text_encoder:175-194 (bidirectional) for rel = key - query in -(length-1) .. length-1 -> int32 [2*length-1]."""
    rel = torch.arange(-(length - 1), length)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    out = out + torch.where(a < max_exact, a, large)
    return out.numpy().astype(np.int32)


def t5_gelu(x: Tensor, q) -> Tensor:
    """text_encoder:16-20, op by op."""
    a = q(0.5 * x)
    p = q(torch.pow(x, 3.0))
    c = q(x + q(0.044715 * p))
    t = q(torch.tanh(q(math.sqrt(2.0 / math.pi) * c)))
    return q(a * q(1.0 + t))


def t5_norm(x: Tensor, w: Tensor, q, eps: float = 1e-6) -> Tensor:
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    return q(w * q(y))


def t5_encode(sd: Dict[str, Tensor], ids: Tensor, n_valid: int, cfg: dict, rounding: Optional[str] = None, max_dist: int = 128) -> Tensor:
    """ids int64 [L], keys = positions < n_valid  ->  [L, dim] fp32 (all rows, as text_encoder(ids, mask) returns them)."""
    q = _r(rounding)
    sd = {k: q(v.float()) for k, v in sd.items()}
    L, n, nl = ids.shape[0], cfg["num_heads"], cfg["num_layers"]
    c = cfg["dim_attn"] // n
    tab = torch.from_numpy(relative_position_buckets(cfg["num_buckets"], max_dist, L)).long()
    idx = tab[(torch.arange(L)[None, :] - torch.arange(L)[:, None]) + L - 1]            # [Lq, Lk]
    x = sd["token_embedding.weight"][ids]
    for i in range(nl):
        b = f"blocks.{i}."
        h = t5_norm(x, sd[b + "norm1.weight"], q)
        qq = q(h @ sd[b + "attn.q.weight"].T).view(L, n, c)
        kk = q(h @ sd[b + "attn.k.weight"].T).view(L, n, c)
        vv = q(h @ sd[b + "attn.v.weight"].T).view(L, n, c)
        emb = sd[("" if cfg.get("shared_pos") else b) + "pos_embedding.embedding.weight"]        # [buckets, heads]
        bias = emb[idx].permute(2, 0, 1)                                                   # [n, Lq, Lk]
        s = q(q(torch.einsum("inc,jnc->nij", qq, kk)) + bias)[:, :, :n_valid]             # masked keys contribute exp(min - max) = 0
        p = q(F.softmax(s.float(), dim=-1))
        a = q(torch.einsum("nij,jnc->inc", p, vv[:n_valid])).reshape(L, n * c)
        x = q(x + q(a @ sd[b + "attn.o.weight"].T))
        h = t5_norm(x, sd[b + "norm2.weight"], q)
        g = t5_gelu(q(h @ sd[b + "ffn.gate.0.weight"].T), q)
        f1 = q(h @ sd[b + "ffn.fc1.weight"].T)
        x = q(x + q(q(f1 * g) @ sd[b + "ffn.fc2.weight"].T))
    return t5_norm(x, sd["norm.weight"], q)


def encode_prompt(sd, ids: Tensor, n_valid: int, cfg: dict, rounding: Optional[str] = None) -> Tensor:
    """prompter:99-112: the encoder output with rows >= n_valid zeroed."""
    out = t5_encode(sd, ids, n_valid, cfg, rounding).clone()
    out[n_valid:] = 0
    return out


# --------------------------------------------------------------------------------------------------------------------- CLIP
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)        # image_encoder:783-784
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(images: Tensor, size: int) -> Tensor:
    """image_encoder:866-873: [B,3,H,W] in [-1,1] -> normalised [B,3,size,size]."""
    x = F.interpolate(images.float(), size=(size, size), mode="bicubic", align_corners=False)
    x = x * 0.5 + 0.5
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (x - mean) / std


def clip_encode_image(sd: Dict[str, Tensor], images: Tensor, cfg: dict, eps: float = 1e-5) -> Tensor:
    """[B,3,H,W] -> [B, tokens, dim] fp32: hidden states after block num_layers-2 (use_31_block)."""
    sd = {k: v.float() for k, v in sd.items()}
    dim, n = cfg["dim"], cfg["num_heads"]
    x = clip_preprocess(images, cfg["image_size"])
    x = F.conv2d(x, sd["patch_embedding.weight"], stride=cfg["patch_size"]).flatten(2).permute(0, 2, 1)
    B = x.shape[0]
    x = torch.cat([sd["cls_embedding"].expand(B, -1, -1), x], dim=1) + sd["pos_embedding"]
    x = F.layer_norm(x, (dim,), sd["pre_norm.weight"], sd["pre_norm.bias"], eps)
    for i in range(cfg["num_layers"] - 1):
        p = f"transformer.{i}."
        h = F.layer_norm(x, (dim,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = h @ sd[p + "attn.to_qkv.weight"].T + sd[p + "attn.to_qkv.bias"]
        q, k, v = (u.view(B, -1, n, dim // n).transpose(1, 2) for u in qkv.chunk(3, dim=-1))
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dim // n)
        a = (F.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, -1, dim)
        x = x + (a @ sd[p + "attn.proj.weight"].T + sd[p + "attn.proj.bias"])
        h = F.layer_norm(x, (dim,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = F.gelu(h @ sd[p + "mlp.0.weight"].T + sd[p + "mlp.0.bias"])
        x = x + (h @ sd[p + "mlp.2.weight"].T + sd[p + "mlp.2.bias"])
    return x
