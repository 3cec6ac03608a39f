"""The MX-fp8 oracle (oracle/mx8_oracle.py) against the published format it restates (OCP Microscaling v1.0, MXFP8 / E4M3): known
answers worked out by hand, and the properties the format guarantees.  CPU only."""
import numpy as np
import torch

from oracle import mx8_oracle as mx


def test_known_answers():
    x = torch.zeros(1, 32)
    x[0, 0], x[0, 1], x[0, 2], x[0, 3] = 1.0, -0.4375, 3.0e-3, 0.99
    q, e = mx.mx8_quantize(x)
    # amax = 1.0 = 2^0 -> shared scale 2^(0 - 8): E8M0 code 127 - 8 = 119; elements x * 2^8
    assert int(e[0, 0]) == 119
    want = torch.tensor([256.0, -112.0, 0.75, 256.0])        # 3e-3 * 256 = 0.768 -> 0.75 (step 0.125 below 1... e4m3 subnormal step 2^-9 above) ; 0.99*256 = 253.44 -> 256 (step 32)
    assert torch.equal(q[0, :4].float(), want)
    back = mx.mx8_dequantize(q, e)
    assert torch.equal(back[0, :4], want / 256.0)
    # a block maximum just below a power of two uses the top of the e4m3 range: 1.99 -> scale 2^-8, 509.44 saturates to 448
    y = torch.zeros(1, 32); y[0, 5] = 1.99
    q2, e2 = mx.mx8_quantize(y)
    assert int(e2[0, 0]) == 119 and float(q2[0, 5].float()) == 448.0
    # zero block: code 0, elements zero
    q3, e3 = mx.mx8_quantize(torch.zeros(2, 64))
    assert not e3.any() and not q3.float().any()


def test_format_properties():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 256, generator=g) * torch.logspace(-6, 4, 37).unsqueeze(1)
    q, e = mx.mx8_quantize(x)
    back = mx.mx8_dequantize(q, e)
    blocks = x.reshape(37, 8, 32)
    amax = blocks.abs().amax(-1)
    scale = torch.ldexp(torch.ones_like(amax), e.int() - 127)
    # the block maximum lands in [2^8, 2^9) in element units: the top binade of e4m3 (448 = 1.75 * 2^8 saturates what lies above)
    top = amax / scale
    assert bool(((top >= 256) & (top < 512)).all())
    # element error: half an e4m3 step of the element's own binade, relative to the block maximum at most 2^-4 (+ the saturation band)
    err = (back - x).abs().reshape(37, 8, 32)
    assert bool((err <= amax[..., None] * 2.0 ** -3).all())
    rel = float((back - x).norm() / x.norm())
    assert rel < 4e-2, rel
    # quantising the dequantised values again is the identity (the format is closed under its own rounding)
    q2, e2 = mx.mx8_quantize(back)
    assert torch.equal(q2.view(torch.uint8), q.view(torch.uint8)) and torch.equal(e2, e)


def test_scale_table_layout():
    e = torch.arange(3 * 8, dtype=torch.uint8).reshape(3, 8) + 100
    t = mx.scale_table(e, 256)
    assert tuple(t.shape) == (2, 256)
    assert int(t[1, 2]) == (100 + 2 * 8 + 4) | ((100 + 2 * 8 + 5) << 8) | ((100 + 2 * 8 + 6) << 16) | ((100 + 2 * 8 + 7) << 24)
    assert not t[:, 3:].any()


def test_linear_matches_dequantised_product():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, 128, generator=g)
    w = torch.randn(7, 128, generator=g).to(torch.float8_e4m3fn).float()
    y = mx.mx8_linear(x, w, torch.ones(7))
    q, e = mx.mx8_quantize(x)
    assert torch.allclose(y, mx.mx8_dequantize(q, e) @ w.t() + 1.0, rtol=1e-6, atol=1e-6)
    assert float((y - (x @ w.t() + 1.0)).norm() / (x @ w.t()).norm()) < 5e-2
