"""Operator seams of the reference, served by the HIP kernels.

Same names and argument meaning as the reference's own functions so that parity tests read like
calls into the reference:
    flash_attention(q, k, v, num_heads)                    models/wan_video_dit.py:116-147
    modulate / LayerNorm  -> layernorm_modulate            models/wan_video_dit.py:150-151,331-333
    RMSNorm + rope_apply  -> rmsnorm_rope                  models/wan_video_dit.py:186-197,178-183
    nn.Linear (+ fused epilogues) -> linear                models/wan_video_dit.py:227-229,242,334-335
    CFG combine + FlowMatchScheduler.step -> cfg_step      pipelines/svi_video.py:410,420
All tensors must be CUDA(HIP) bf16 and contiguous unless noted.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L


def _chk(t: torch.Tensor, name: str, dtype=torch.bfloat16):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (svi_hip has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int,
                    compatibility_mode: bool = False) -> torch.Tensor:
    """[b, s, (n d)] layout in and out; d = 128."""
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    b, sq, dim = q.shape
    skv = k.shape[1]
    out = torch.empty_like(q)
    L.check(L.lib().svi_attention_fwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), b, sq, skv, num_heads,
                                      dim // num_heads, L.current_stream()), "flash_attention")
    return out


def fp8_attention(enable: bool = True) -> None:
    """Opt-in, process-wide, NOT the reference's arithmetic: every long-sequence attention call (flash_attention above, the DiT's self-attention) quantises
    Q and K to MX e4m3 and takes QK^T on the scaled fp8 matrix path; softmax and P·V stay as they are.  The reference's own dispatch offers this place to a
    quantised-QK^T backend (models/wan_video_dit.py:116-147, SageAttention).  Same as SVI_ATTN_QK8=1 in the environment; loops that hold a captured step graph
    re-capture.  Separately toleranced: tests/test_gpu_attn_qk8.py."""
    L.set_switch("SVI_ATTN_QK8", 1 if enable else None)


def fp8_attention_enabled() -> bool:
    """What the LIBRARY parsed (svi_switch_state), not the environment of the moment: an environment edit without a reload changes nothing the kernels do."""
    return L.lib().svi_switch_state(b"SVI_ATTN_QK8") == 1


def layernorm_modulate(x: torch.Tensor, eps: float = 1e-6, weight: Optional[torch.Tensor] = None,
                       bias: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
                       scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm over the last dim [+ affine] [+ x*(1+scale)+shift]; shift/scale are [dim] vectors."""
    _chk(x, "x")
    dim = x.shape[-1]
    rows = x.numel() // dim
    out = torch.empty_like(x)
    for t, n in ((weight, "weight"), (bias, "bias"), (shift, "shift"), (scale, "scale")):
        if t is not None:
            _chk(t, n)
            if t.numel() != dim:
                raise ValueError(f"{n} must have {dim} elements")
    L.check(L.lib().svi_layernorm_modulate(L.ptr(x), L.ptr(out), rows, dim, eps, L.ptr(weight), L.ptr(bias),
                                           L.ptr(shift), L.ptr(scale), L.current_stream()), "layernorm_modulate")
    return out


def rmsnorm_rope_(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, grid=None, num_heads: int = 0) -> torch.Tensor:
    """In place on x [rows, dim]: RMSNorm over dim, then 3-D RoPE for the (f, h, w) token grid if given."""
    _chk(x, "x"); _chk(weight, "weight")
    rows, dim = x.shape[-2], x.shape[-1]
    f, h, w = grid if grid is not None else (0, 0, 0)
    L.check(L.lib().svi_rmsnorm_rope(L.ptr(x), dim, rows, dim, L.ptr(weight), eps, 1 if grid is not None else 0,
                                      num_heads, f, h, w, L.current_stream()), "rmsnorm_rope")
    return x


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = L.EPI_BIAS,
           gate: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           transpose_out: bool = False) -> torch.Tensor:
    """y = epilogue(x @ weight.T + bias).  x [M, K], weight [N, K].  transpose_out returns y^T [N, M8]
    (M8 = M rounded up to 8; the form the attention kernel wants V in)."""
    _chk(x, "x"); _chk(weight, "weight")
    M, K = x.shape
    N = weight.shape[0]
    if bias is not None:
        _chk(bias, "bias")
    if gate is not None:
        _chk(gate, "gate", torch.float32)
    st = L.current_stream()
    if transpose_out:
        ld = (M + 7) // 8 * 8
        out = torch.zeros((N, ld), dtype=torch.bfloat16, device=x.device)
        L.check(L.lib().svi_gemm_bf16(L.ptr(weight), K, L.ptr(x), K, L.ptr(out), ld, N, M, K, L.ptr(bias), 1,
                                      epilogue, None, None, 0, st), "linear^T")
        return out
    ldc = (N + 7) // 8 * 8
    out = torch.empty((M, ldc), dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        _chk(residual, "residual")
    L.check(L.lib().svi_gemm_bf16(L.ptr(x), K, L.ptr(weight), K, L.ptr(out), ldc, M, N, K, L.ptr(bias), 0, epilogue,
                                  L.ptr(gate), L.ptr(residual), residual.shape[-1] if residual is not None else 0, st),
            "linear")
    return out if ldc == N else out[:, :N].contiguous()


def cfg_step_(latents: torch.Tensor, cond: torch.Tensor, uncond: Optional[torch.Tensor], cfg_scale: float,
              dsigma: float) -> torch.Tensor:
    """latents += (uncond + cfg_scale*(cond-uncond)) * dsigma, in place, bf16 op-by-op rounding."""
    _chk(latents, "latents"); _chk(cond, "cond")
    if uncond is not None:
        _chk(uncond, "uncond")
    L.check(L.lib().svi_cfg_step(L.ptr(latents), L.ptr(cond), L.ptr(uncond), latents.numel(), float(cfg_scale),
                                 float(dsigma), L.current_stream()), "cfg_step")
    return latents


def fp8_e4m3_to_bf16(t: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a float8_e4m3fn tensor (exact): the cast the reference's FP8 mode performs in front of every use
    (vram_management/layers.py:65-71), done once."""
    if t.dtype != torch.float8_e4m3fn:
        raise ValueError(f"expected a float8_e4m3fn tensor, got {t.dtype}")
    src = t.to(device="cuda").contiguous()
    out = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    L.check(L.lib().svi_fp8_e4m3_to_bf16(L.ptr(src.view(torch.uint8)), L.ptr(out), src.numel(), L.current_stream()), "fp8_e4m3_to_bf16")
    return out


def cfg3_step_(latents: torch.Tensor, cond: torch.Tensor, uncond: torch.Tensor, drop_text: torch.Tensor, text_scale: float, audio_scale: float,
               dsigma: float) -> torch.Tensor:
    """latents += (uncond + text_scale*(cond - drop_text) + audio_scale*(drop_text - uncond)) * dsigma, in place, bf16 op-by-op
    (the talk sampler's guidance, svi_video_talk.py:455-461)."""
    for t, n in ((latents, "latents"), (cond, "cond"), (uncond, "uncond"), (drop_text, "drop_text")):
        _chk(t, n)
    L.check(L.lib().svi_cfg3_step(L.ptr(latents), L.ptr(cond), L.ptr(uncond), L.ptr(drop_text), latents.numel(), float(text_scale),
                                  float(audio_scale), float(dsigma), L.current_stream()), "cfg3_step")
    return latents


def linear_row_stats(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, eps: float = 1e-6):
    """nn.Linear whose epilogue also leaves RMSNorm's statistic of the OUTPUT rows (svi_linear_row_stats): returns (y bf16 [M, N], rs fp32 [M] =
    rsqrt(mean(y^2) + eps), row_sumsq fp32 [N/64, M]).  The cross-attention query projection of the DiT block (dit:296) runs this way."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    if bias is not None:
        bias = _chk(bias, "bias")
    M, K = x.shape
    N = weight.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    ss = torch.empty((N // 64, M), dtype=torch.float32, device=x.device)
    rs = torch.empty((M,), dtype=torch.float32, device=x.device)
    L.check(L.lib().svi_linear_row_stats(L.ptr(x), K, L.ptr(weight), K, L.ptr(y), N, M, N, K, L.ptr(bias), float(eps), L.ptr(ss), M, L.ptr(rs),
                                         L.current_stream()), "linear_row_stats")
    return y, rs, ss


def cross_attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, num_heads: int, s_kv: Optional[int] = None, q_rs: Optional[torch.Tensor] = None,
                    q_gain: Optional[torch.Tensor] = None, q_out_scale: float = 1.0, key_tail: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The DiT block's cross-attention over the prompt's (short) key axis (svi_cross_attention_fwd): q bf16 [Lq, n*128] — with q_rs / q_gain the RAW
    projection output, RMS-normalised and scaled as it is read; without them used as it is (it must carry softmax_scale*log2e) — k bf16 [Lk, n*128]
    normalised, vt bf16 [n*128, ldvt] = V transposed (columns >= Lk up to a multiple of 8 readable)."""
    q, k, vt = _chk(q, "q"), _chk(k, "k"), _chk(vt, "vt")
    Lq, Lk = q.shape[0], (s_kv if s_kv is not None else k.shape[0])
    out = torch.empty((Lq, num_heads * 128), dtype=torch.bfloat16, device=q.device)
    if q_rs is not None:
        q_rs, q_gain = _chk(q_rs, "q_rs", torch.float32), _chk(q_gain, "q_gain")
    if key_tail is not None:
        key_tail = _chk(key_tail, "key_tail", torch.int32)
    L.check(L.lib().svi_cross_attention_fwd(L.ptr(q), q.shape[1], L.ptr(k), k.shape[1], L.ptr(vt), vt.shape[1], L.ptr(out), out.shape[1], Lq, Lk, num_heads,
                                            L.ptr(key_tail), L.ptr(q_rs), L.ptr(q_gain), float(q_out_scale), L.current_stream()), "cross_attention")
    return out
