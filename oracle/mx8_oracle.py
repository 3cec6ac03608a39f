"""CPU restatement of the opt-in MX-fp8 MLP and projection paths of the HIP backend (csrc/svi_gemm.hip mx8_quantize_kernel / gemm_mx8_nt_256_kernel).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).  There is no reference counterpart to pin this to — the reference
computes in bf16 and only STORES weights as float8_e4m3fn (test_svi.py:337, diffsynth/vram_management/layers.py:65-71) — so this file
restates a published format instead: OCP Microscaling (MX) v1.0, MXFP8 with E4M3 elements: blocks of 32 consecutive K elements share
one E8M0 scale X = 2^(floor(log2(max|v|)) - emax_elem), emax_elem(E4M3) = 8; elements are v / X rounded to nearest even, saturated to
+-448.  "Parity" for this path means: the HIP quantiser reproduces these bits, the HIP GEMM equals the dequantised product, and the
resulting DiT output stays within a STATED distance of the bf16 path (tests/test_gpu_mx8.py) — not parity with the reference.
"""
from __future__ import annotations

import torch
from torch import Tensor

BLOCK = 32


def mx8_quantize(x: Tensor):
    """x [R, K] fp32 (K % 32 == 0) -> (q float8_e4m3fn [R, K], E uint8 [R, K / 32]) with value ~= q * 2^(E - 127)."""
    r, k = x.shape
    xb = x.to(torch.float32).reshape(r, k // BLOCK, BLOCK)
    amax = xb.abs().amax(dim=-1)
    eb = (amax.contiguous().view(torch.int32) >> 23) & 0xFF              # biased exponent: floor(log2(amax)) + 127 (0 for zero / subnormal)
    e = torch.clamp(eb - 8, min=0)                                       # E8M0 code of 2^(floor(log2 amax) - 8)
    inv = torch.ldexp(torch.ones_like(amax), 127 - e)                    # 2^(127 - E), exact
    v = torch.clamp(xb * inv[..., None], -448.0, 448.0)
    q = v.reshape(r, k).to(torch.float8_e4m3fn)                          # round to nearest even
    return q, e.to(torch.uint8)


def mx8_dequantize(q: Tensor, e: Tensor) -> Tensor:
    r, k = q.shape
    scale = torch.ldexp(torch.ones(e.shape, dtype=torch.float32), e.to(torch.int32) - 127)
    return (q.to(torch.float32).reshape(r, k // BLOCK, BLOCK) * scale[..., None]).reshape(r, k)


def mx8_linear(x: Tensor, w: Tensor, b=None) -> Tensor:
    """nn.Linear on the MX-fp8 path: activations quantised per row and 32-block, weights e4m3 values with unit scale (`w` holds
    e4m3-representable numbers: the reference's FP8 storage mode), products and sums in fp64 -> fp32."""
    lead = x.shape[:-1]
    q, e = mx8_quantize(x.reshape(-1, x.shape[-1]))
    y = (mx8_dequantize(q, e).double() @ w.double().t()).float().reshape(*lead, w.shape[0])
    return y if b is None else y + b


def scale_table(e: Tensor, sc_rows: int) -> Tensor:
    """E [R, K/32] -> the device layout int32 [K/128][sc_rows]: byte b of dword [kt][m] = block 4 kt + b of row m."""
    r, nb = e.shape
    t = e.to(torch.int64).reshape(r, nb // 4, 4)
    d = t[..., 0] | (t[..., 1] << 8) | (t[..., 2] << 16) | (t[..., 3] << 24)
    out = torch.zeros((nb // 4, sc_rows), dtype=torch.int64)
    out[:, :r] = d.t()
    return out


def mx8_linear_t(w: Tensor, x: Tensor, b=None) -> Tensor:
    """The transposed projection C[M, N] = w[M, K] · dequant(quant(x[N, K]))^T + b[:, None] (svi_gemm_mx8_wscaled: the block scales belong to the W operand's rows)."""
    q, e = mx8_quantize(x)
    y = (w.double() @ mx8_dequantize(q, e).double().t()).float()
    return y if b is None else y + b[:, None]
