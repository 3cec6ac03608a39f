"""CPU oracle for the Wan DiT denoiser forward — TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain functional torch-CPU fp32/fp64
arithmetic, of what the reference computes on the rolling-window denoising hot
path.  It is the checker for the HIP path; nothing in the product package
(`stable-video-infinity_amd/`) imports it.  Only `tests/`, `bench.py`'s
`cpu_baseline` leg and `__graft_entry__.smoke()` may call it.

Parity status: PINNED.  `tests/golden/dit_*.npz` hold outputs of the reference's
own `WanModel.forward` / `DiTBlock.forward` (imported from /root/reference by
`tests/gen_golden.py`, committed) on seeded inputs; `tests/test_oracle_dit.py`
checks this oracle against them to fp32 round-off.

Reference locations restated here (all under diffsynth/):
  models/wan_video_dit.py:150-151  modulate
  models/wan_video_dit.py:154-158  sinusoidal_embedding_1d
  models/wan_video_dit.py:161-175  precompute_freqs_cis(_3d)
  models/wan_video_dit.py:178-183  rope_apply
  models/wan_video_dit.py:186-197  RMSNorm
  models/wan_video_dit.py:210-242  SelfAttention
  models/wan_video_dit.py:245-303  CrossAttention (text branch + CLIP image branch)
  models/wan_video_dit.py:354-374  DiTBlock.forward
  models/wan_video_dit.py:392-404  Head
  models/wan_video_dit.py:473-484  patchify / unpatchify
  pipelines/svi_video.py:74-137    model_fn_wan_video (TeaCache/USP hooks excluded)

Two arithmetic modes:
  * rounding=None       every op in fp32 (RoPE and the timestep sinusoid in fp64,
                        as the reference does regardless of model dtype);
  * rounding="bf16"     fp32 math, but every tensor the reference would have
                        materialised as a bf16 tensor is rounded to bf16 at that
                        point.  This is the numerical contract of the HIP path
                        (bf16 storage, fp32 accumulate) and gives the tight
                        comparison; the fp32 mode gives the independent one.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor


@dataclass(frozen=True)
class DiTConfig:
    """Constructor arguments of the reference WanModel (wan_video_dit.py:408-421)."""
    dim: int = 1536
    in_dim: int = 16
    ffn_dim: int = 8960
    out_dim: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    eps: float = 1e-6
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    num_heads: int = 12
    num_layers: int = 30
    has_image_input: bool = False
    enable_multitalk: bool = False

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads


WAN_1_3B_T2V = DiTConfig()
WAN_14B_I2V = DiTConfig(dim=5120, in_dim=36, ffn_dim=13824, out_dim=16, num_heads=40,
                        num_layers=40, has_image_input=True)


def _rounder(rounding: Optional[str]):
    if rounding is None:
        return lambda t: t
    if rounding == "bf16":
        return lambda t: t.to(torch.bfloat16).to(torch.float32)
    raise ValueError(f"unknown rounding mode {rounding!r}")


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def timestep_sinusoid(freq_dim: int, timestep: Tensor) -> Tensor:
    """[B] -> [B, freq_dim]; cos half then sin half, computed in fp64 (dit:154-158)."""
    half = freq_dim // 2
    expo = torch.arange(half, dtype=torch.float64) / half
    ang = timestep.to(torch.float64)[:, None] * torch.pow(torch.tensor(10000.0, dtype=torch.float64), -expo)[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1).to(timestep.dtype)


def rope_axis_table(axis_dim: int, length: int, theta: float = 10000.0) -> Tensor:
    """complex128 [length, axis_dim//2] rotation table for one axis (dit:169-175)."""
    inv = 1.0 / (theta ** (torch.arange(0, axis_dim, 2)[: axis_dim // 2].double() / axis_dim))
    ang = torch.outer(torch.arange(length, dtype=torch.float64), inv)
    return torch.polar(torch.ones_like(ang), ang)


def rope_table_3d(head_dim: int, grid: Tuple[int, int, int]) -> Tensor:
    """complex128 [f*h*w, head_dim//2]: frame | height | width frequency bands (dit:161-166 +
    the gather at svi_video.py:106-110).  head_dim=128 -> 22 + 21 + 21 complex pairs."""
    f, h, w = grid
    d_hw = head_dim // 3
    d_f = head_dim - 2 * d_hw
    tf, th, tw = rope_axis_table(d_f, f), rope_axis_table(d_hw, h), rope_axis_table(d_hw, w)
    if tf.shape[1] + th.shape[1] + tw.shape[1] != head_dim // 2:
        raise ValueError("head_dim does not split into even frame/height/width bands (needs e.g. 128)")
    tab = torch.cat([
        tf[:, None, None, :].expand(f, h, w, -1),
        th[None, :, None, :].expand(f, h, w, -1),
        tw[None, None, :, :].expand(f, h, w, -1)], dim=-1)
    return tab.reshape(f * h * w, head_dim // 2)


def apply_rope(x: Tensor, table: Tensor, num_heads: int) -> Tensor:
    """x [B,L,H*dh] fp32; adjacent element pairs are (re,im); fp64 complex multiply (dit:178-183)."""
    b, l, d = x.shape
    xc = torch.view_as_complex(x.to(torch.float64).reshape(b, l, num_heads, d // num_heads // 2, 2))
    out = torch.view_as_real(xc * table[None, :, None, :]).reshape(b, l, d)
    return out.to(torch.float32)


def rms_norm_full(x: Tensor, weight: Tensor, eps: float, rnd) -> Tensor:
    """RMS over the whole model dim (all heads jointly), fp32 math; the weight multiply
    happens after the cast back to the model dtype (dit:192-197)."""
    y = x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)
    return rnd(rnd(y) * weight)


def layer_norm(x: Tensor, eps: float, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def gelu_tanh(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def attention(q: Tensor, k: Tensor, v: Tensor, num_heads: int, q_chunk: int = 512) -> Tensor:
    """[B,Lq,H*dh] x [B,Lk,H*dh] -> [B,Lq,H*dh]; unmasked softmax(QK^T/sqrt(dh))V (dit:116-147).
    Queries are processed `q_chunk` rows at a time (exact; only bounds the size of the score matrix)."""
    b, lq, d = q.shape
    dh = d // num_heads
    qh = q.reshape(b, lq, num_heads, dh).permute(0, 2, 1, 3)
    kt = k.reshape(b, -1, num_heads, dh).permute(0, 2, 3, 1)
    vh = v.reshape(b, -1, num_heads, dh).permute(0, 2, 1, 3)
    out = torch.empty_like(qh)
    lk = kt.shape[-1]
    q_chunk = max(1, min(q_chunk, lq))
    sc = torch.empty((b, num_heads, q_chunk, lk), dtype=q.dtype)          # reused: avoids multi-GB realloc per chunk
    scale = 1.0 / math.sqrt(dh)
    for s0 in range(0, lq, q_chunk):
        n = min(q_chunk, lq - s0)
        s_ = sc[:, :, :n]
        torch.matmul(qh[:, :, s0:s0 + n], kt, out=s_)
        s_.mul_(scale)
        s_.sub_(s_.amax(dim=-1, keepdim=True)).exp_()
        s_.div_(s_.sum(dim=-1, keepdim=True))
        torch.matmul(s_, vh, out=out[:, :, s0:s0 + n])
    return out.permute(0, 2, 1, 3).reshape(b, lq, d)


# --------------------------------------------------------------------------------------
# block
# --------------------------------------------------------------------------------------
def modulated_norm(x: Tensor, shift: Tensor, scale: Tensor, eps: float, rnd) -> Tensor:
    """LN (no affine) then x*(1+scale)+shift with the reference's op-by-op rounding (dit:150,358)."""
    xn = rnd(layer_norm(x, eps))
    return rnd(rnd(xn * rnd(1.0 + scale)) + shift)


def self_attention(sd: Dict[str, Tensor], p: str, x: Tensor, rope: Tensor, cfg: DiTConfig, rnd, proj=None) -> Tensor:
    """proj: the linear layer the token-side projections run on (default: `linear`; the HIP path's opt-in MX-fp8 projections are checked with
    oracle/mx8_oracle.mx8_linear here, as dit_block's mlp_linear does for the MLP)."""
    lin = proj or linear
    q = rms_norm_full(rnd(lin(x, sd[p + "q.weight"], sd[p + "q.bias"])), sd[p + "norm_q.weight"], cfg.eps, rnd)
    k = rms_norm_full(rnd(lin(x, sd[p + "k.weight"], sd[p + "k.bias"])), sd[p + "norm_k.weight"], cfg.eps, rnd)
    v = rnd(lin(x, sd[p + "v.weight"], sd[p + "v.bias"]))
    q = rnd(apply_rope(q, rope, cfg.num_heads))
    k = rnd(apply_rope(k, rope, cfg.num_heads))
    a = rnd(attention(q, k, v, cfg.num_heads))
    return rnd(lin(a, sd[p + "o.weight"], sd[p + "o.bias"]))


def cross_attention(sd: Dict[str, Tensor], p: str, x: Tensor, context: Tensor, cfg: DiTConfig, rnd, proj=None) -> Tensor:
    """proj: as in self_attention, for the two token-side projections (q and o); the prompt-side k / v stay on `linear`."""
    lin = proj or linear
    if cfg.has_image_input:
        img, ctx = context[:, :257], context[:, 257:]
    else:
        img, ctx = None, context
    q = rms_norm_full(rnd(lin(x, sd[p + "q.weight"], sd[p + "q.bias"])), sd[p + "norm_q.weight"], cfg.eps, rnd)
    k = rms_norm_full(rnd(linear(ctx, sd[p + "k.weight"], sd[p + "k.bias"])), sd[p + "norm_k.weight"], cfg.eps, rnd)
    v = rnd(linear(ctx, sd[p + "v.weight"], sd[p + "v.bias"]))
    a = rnd(attention(q, k, v, cfg.num_heads))
    if img is not None:
        ki = rms_norm_full(rnd(linear(img, sd[p + "k_img.weight"], sd[p + "k_img.bias"])),
                           sd[p + "norm_k_img.weight"], cfg.eps, rnd)
        vi = rnd(linear(img, sd[p + "v_img.weight"], sd[p + "v_img.bias"]))
        a = rnd(a + rnd(attention(q, ki, vi, cfg.num_heads)))
    return rnd(lin(a, sd[p + "o.weight"], sd[p + "o.bias"]))


def audio_tokens(sd: Dict[str, Tensor], audio_first: Tensor, audio_latter: Tensor, rnd) -> Tensor:
    """AudioProjModel.forward (dit:82-115) for one clip: audio_first [1, 5, 12, 768], audio_latter [f-1, 8, 12, 768] ->
    [f, 32, 768] context tokens (LayerNorm(768) applied: norm_output_audio=True, dit:461)."""
    p = "audio_proj."
    h0 = rnd(torch.relu(rnd(linear(audio_first.reshape(audio_first.shape[0], -1), sd[p + "proj1.weight"], sd[p + "proj1.bias"]))))
    h1 = rnd(torch.relu(rnd(linear(audio_latter.reshape(audio_latter.shape[0], -1), sd[p + "proj1_vf.weight"], sd[p + "proj1_vf.bias"]))))
    h = torch.cat([h0, h1], dim=0)
    h = rnd(torch.relu(rnd(linear(h, sd[p + "proj2.weight"], sd[p + "proj2.bias"]))))
    tok = rnd(linear(h, sd[p + "proj3.weight"], sd[p + "proj3.bias"])).reshape(h.shape[0], 32, 768)
    return rnd(layer_norm(tok, 1e-5, sd[p + "norm.weight"], sd[p + "norm.bias"]))


def audio_cross_attention(sd: Dict[str, Tensor], prefix: str, x: Tensor, audio: Tensor, frames: int, cfg: DiTConfig, rnd) -> Tensor:
    """x_a of DiTBlock.forward (dit:361-366): SingleStreamAttention.forward with human_num == 1 (models/attention.py:318-371) on
    norm_x(x): every frame's tokens attend to that frame's 32 audio tokens; no q/k norm, no RoPE, scale head_dim^-0.5.
    x [1, L, D], audio [f, 32, 768]."""
    p = prefix + "audio_cross_attn."
    D = cfg.dim
    xn = rnd(layer_norm(x, cfg.eps, sd[prefix + "norm_x.weight"], sd[prefix + "norm_x.bias"]))          # WanLayerNorm: fp32, one rounding
    q = rnd(linear(xn, sd[p + "q_linear.weight"], sd[p + "q_linear.bias"])).reshape(frames, -1, D)       # '(B N_t) S C'
    kv = rnd(linear(audio, sd[p + "kv_linear.weight"], sd[p + "kv_linear.bias"]))                       # [f, 32, 2D]: (2, H, hd) split
    k, v = kv[..., :D], kv[..., D:]
    a = rnd(attention(q, k, v, cfg.num_heads)).reshape(1, -1, D)
    return rnd(linear(a, sd[p + "proj.weight"], sd[p + "proj.bias"]))


def dit_block(sd: Dict[str, Tensor], prefix: str, x: Tensor, context: Tensor, t_mod: Tensor, rope: Tensor,
              cfg: DiTConfig, rounding: Optional[str] = None, audio: Optional[Tensor] = None, frames: int = 0, mlp_linear=None, proj_linear=None) -> Tensor:
    """One DiTBlock (dit:354-374).  x [B,L,D], context [B,Lc,D] (already text-embedded),
    t_mod [B,6,D], rope complex128 [L, dh/2]; audio [f, 32, 768] (talk variant) or None."""
    rnd = _rounder(rounding)
    mod = rnd(sd[prefix + "modulation"] + t_mod)                       # [B,6,D]
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = [mod[:, i:i + 1] for i in range(6)]
    h = modulated_norm(x, sh_a, sc_a, cfg.eps, rnd)
    x = rnd(x + rnd(g_a * self_attention(sd, prefix + "self_attn.", h, rope, cfg, rnd, proj_linear)))
    h = rnd(layer_norm(x, cfg.eps, sd[prefix + "norm3.weight"], sd[prefix + "norm3.bias"]))
    x = rnd(x + cross_attention(sd, prefix + "cross_attn.", h, context, cfg, rnd, proj_linear))
    if audio is not None:
        x = rnd(x + audio_cross_attention(sd, prefix, x, audio, frames, cfg, rnd))                     # dit:364-366
    h = modulated_norm(x, sh_m, sc_m, cfg.eps, rnd)
    lin = mlp_linear or linear          # the opt-in MX-fp8 MLP of the HIP path is checked with oracle/mx8_oracle.mx8_linear here
    u = rnd(gelu_tanh(rnd(lin(h, sd[prefix + "ffn.0.weight"], sd[prefix + "ffn.0.bias"]))))
    x = rnd(x + rnd(g_m * rnd(lin(u, sd[prefix + "ffn.2.weight"], sd[prefix + "ffn.2.bias"]))))
    return x


def dit_block_rows(sd: Dict[str, Tensor], prefix: str, x: Tensor, context: Tensor, t_mod: Tensor, rope: Tensor, cfg: DiTConfig,
                   rows, rounding: Optional[str] = None) -> Tensor:
    """dit_block's output at the token rows `rows` only — exactly those rows of dit_block(...) (the same statements on fewer rows): self-attention
    needs K and V of every token, everything else of a block is row-local.  What makes a block at sizes beyond the host's reach for the whole
    score matrix checkable (81f@1280x720: 75600 tokens)."""
    rnd = _rounder(rounding)
    rows = torch.as_tensor(rows, dtype=torch.long)
    mod = rnd(sd[prefix + "modulation"] + t_mod)
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = [mod[:, i:i + 1] for i in range(6)]
    h = modulated_norm(x, sh_a, sc_a, cfg.eps, rnd)
    p = prefix + "self_attn."
    k = rms_norm_full(rnd(linear(h, sd[p + "k.weight"], sd[p + "k.bias"])), sd[p + "norm_k.weight"], cfg.eps, rnd)
    k = rnd(apply_rope(k, rope, cfg.num_heads))
    v = rnd(linear(h, sd[p + "v.weight"], sd[p + "v.bias"]))
    q = rms_norm_full(rnd(linear(h[:, rows], sd[p + "q.weight"], sd[p + "q.bias"])), sd[p + "norm_q.weight"], cfg.eps, rnd)
    q = rnd(apply_rope(q, rope[rows], cfg.num_heads))
    a = rnd(attention(q, k, v, cfg.num_heads))
    xr = rnd(x[:, rows] + rnd(g_a * rnd(linear(a, sd[p + "o.weight"], sd[p + "o.bias"]))))
    h = rnd(layer_norm(xr, cfg.eps, sd[prefix + "norm3.weight"], sd[prefix + "norm3.bias"]))
    xr = rnd(xr + cross_attention(sd, prefix + "cross_attn.", h, context, cfg, rnd))
    h = modulated_norm(xr, sh_m, sc_m, cfg.eps, rnd)
    u = rnd(gelu_tanh(rnd(linear(h, sd[prefix + "ffn.0.weight"], sd[prefix + "ffn.0.bias"]))))
    return rnd(xr + rnd(g_m * rnd(linear(u, sd[prefix + "ffn.2.weight"], sd[prefix + "ffn.2.bias"]))))


# --------------------------------------------------------------------------------------
# whole forward
# --------------------------------------------------------------------------------------
def patchify(x: Tensor, w: Tensor, b: Tensor, patch: Tuple[int, int, int]) -> Tuple[Tensor, Tuple[int, int, int]]:
    """Non-overlapping patch projection == Conv3d(kernel=stride=patch); tokens ordered (f h w) (dit:473-477)."""
    bsz, c, t, hh, ww = x.shape
    pt, ph, pw = patch
    f, h, wd = t // pt, hh // ph, ww // pw
    cols = x.reshape(bsz, c, f, pt, h, ph, wd, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(bsz, f * h * wd, c * pt * ph * pw)
    return cols @ w.reshape(w.shape[0], -1).t() + b, (f, h, wd)


def unpatchify(tok: Tensor, grid: Tuple[int, int, int], patch: Tuple[int, int, int], out_dim: int) -> Tensor:
    """[B,(f h w),(pt ph pw c)] -> [B,c,f*pt,h*ph,w*pw] (dit:479-484)."""
    f, h, w = grid
    pt, ph, pw = patch
    b = tok.shape[0]
    return tok.reshape(b, f, h, w, pt, ph, pw, out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(
        b, out_dim, f * pt, h * ph, w * pw)


def embed_time(sd: Dict[str, Tensor], cfg: DiTConfig, timestep: Tensor, rnd) -> Tuple[Tensor, Tensor]:
    """-> (t [B,D], t_mod [B,6,D]) (svi_video.py:92-93)."""
    e = rnd(timestep_sinusoid(cfg.freq_dim, timestep.to(torch.float32)))
    h = rnd(linear(e, sd["time_embedding.0.weight"], sd["time_embedding.0.bias"]))
    h = rnd(h * torch.sigmoid(h))
    t = rnd(linear(h, sd["time_embedding.2.weight"], sd["time_embedding.2.bias"]))
    s = rnd(t * torch.sigmoid(t))
    t_mod = rnd(linear(s, sd["time_projection.1.weight"], sd["time_projection.1.bias"])).unflatten(1, (6, cfg.dim))
    return t, t_mod


def embed_text(sd: Dict[str, Tensor], context: Tensor, rnd) -> Tensor:
    h = rnd(gelu_tanh(rnd(linear(context, sd["text_embedding.0.weight"], sd["text_embedding.0.bias"]))))
    return rnd(linear(h, sd["text_embedding.2.weight"], sd["text_embedding.2.bias"]))


def embed_image(sd: Dict[str, Tensor], clip_feature: Tensor, rnd) -> Tensor:
    """img_emb MLP: LN -> Linear -> GELU(erf) -> Linear -> LN (dit:377-389); I2V models only."""
    p = "img_emb.proj."
    h = rnd(layer_norm(clip_feature, 1e-5, sd[p + "0.weight"], sd[p + "0.bias"]))
    h = rnd(linear(h, sd[p + "1.weight"], sd[p + "1.bias"]))
    h = rnd(0.5 * h * (1.0 + torch.erf(h / math.sqrt(2.0))))
    h = rnd(linear(h, sd[p + "3.weight"], sd[p + "3.bias"]))
    return rnd(layer_norm(h, 1e-5, sd[p + "4.weight"], sd[p + "4.bias"]))


def head(sd: Dict[str, Tensor], cfg: DiTConfig, x: Tensor, t: Tensor, rnd) -> Tensor:
    mod = rnd(sd["head.modulation"] + t[:, None, :])                   # [B,2,D]
    shift, scale = mod[:, 0:1], mod[:, 1:2]
    xn = rnd(layer_norm(x, cfg.eps))
    h = rnd(rnd(xn * rnd(1.0 + scale)) + shift)
    return rnd(linear(h, sd["head.head.weight"], sd["head.head.bias"]))


def dit_forward(sd: Dict[str, Tensor], cfg: DiTConfig, x: Tensor, timestep: Tensor, context: Tensor,
                clip_feature: Optional[Tensor] = None, y: Optional[Tensor] = None,
                add_condition: Optional[Tensor] = None, rounding: Optional[str] = None,
                return_tokens: bool = False, audio_embed_tuple=None) -> Tensor:
    """Velocity prediction for latents x [B,C,T,H,W]; mirrors model_fn_wan_video (svi_video.py:74-137) and, with
    audio_embed_tuple = (first [1,1,5,12,768], latter [1,f-1,8,12,768]), model_fn_wan_talk_video (svi_video_talk.py:83-160).

    sd holds fp32 tensors keyed by the reference state-dict names.  In "bf16" rounding mode the
    caller is expected to pass weights already rounded to bf16 values (as the bf16 model holds)."""
    rnd = _rounder(rounding)
    x = rnd(x.to(torch.float32))
    t, t_mod = embed_time(sd, cfg, timestep, rnd)
    ctx = embed_text(sd, rnd(context.to(torch.float32)), rnd)
    if cfg.has_image_input:
        x = torch.cat([x, rnd(y.to(torch.float32))], dim=1)
        ctx = torch.cat([embed_image(sd, rnd(clip_feature.to(torch.float32)), rnd), ctx], dim=1)
    tok, grid = patchify(x, sd["patch_embedding.weight"], sd["patch_embedding.bias"], cfg.patch_size)
    tok = rnd(tok)
    if add_condition is not None:
        tok = rnd(add_condition + tok)
    rope = rope_table_3d(cfg.head_dim, grid)
    audio = None
    if audio_embed_tuple is not None:
        audio = audio_tokens(sd, rnd(audio_embed_tuple[0][0].to(torch.float32)), rnd(audio_embed_tuple[1][0].to(torch.float32)), rnd)
    for i in range(cfg.num_layers):
        tok = dit_block(sd, f"blocks.{i}.", tok, ctx, t_mod, rope, cfg, rounding, audio, grid[0])
    if return_tokens:
        return tok
    out = head(sd, cfg, tok, t, rnd)
    return unpatchify(out, grid, cfg.patch_size, cfg.out_dim)
