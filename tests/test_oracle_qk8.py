"""CPU (not gpu): what the opt-in quantised-QK^T attention (SVI_ATTN_QK8; tests/test_gpu_attn_qk8.py) is checked AGAINST, restated without a device —
attention over MX e4m3 Q and K (oracle/mx8_oracle.py: one E8M0 scale per 32 channels of a head) in fp64.  No reference counterpart exists for this
arithmetic (the reference's attention is bf16; its dispatch, models/wan_video_dit.py:116-147, only accepts a quantised-QK^T backend), so these tests pin
properties of the format the GPU tests rely on, and the size of the effect the mode's stated tolerance is about."""
import numpy as np
import torch

from oracle import mx8_oracle as mx


def attention64(q, k, v, heads):
    L_, D = q.shape
    qh = q.double().view(L_, heads, 128).transpose(0, 1)
    kh = k.double().view(-1, heads, 128).transpose(0, 1)
    vh = v.double().view(-1, heads, 128).transpose(0, 1)
    p = torch.softmax(qh @ kh.transpose(1, 2) / 128 ** 0.5, dim=-1)
    return (p @ vh).transpose(0, 1).reshape(L_, D)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def operands(seed, Lq, Lk, heads, spread):
    g = torch.Generator().manual_seed(seed)
    D = heads * 128
    mag = torch.exp(torch.randn((D,), generator=g) * spread)
    q = (torch.randn((Lq, D), generator=g) * mag).to(torch.bfloat16).float()
    k = (torch.randn((Lk, D), generator=g) * mag.flip(0)).to(torch.bfloat16).float()
    v = torch.randn((Lk, D), generator=g).to(torch.bfloat16).float()
    return q, k, v


def test_blocks_are_per_head_and_scales_are_powers_of_two():
    """A head's 128 channels are four MX blocks: the scale table of a [L, heads * 128] operand is one dword per (head, token) — what the kernel fetches."""
    q, _, _ = operands(1, 37, 8, 3, 1.0)
    e8, e = mx.mx8_quantize(q)
    assert e.shape == (37, 3 * 4) and e.dtype == torch.uint8
    tab = mx.scale_table(e, 40)
    assert tab.shape == (3, 40)
    for h in range(3):
        for b in range(4):
            assert torch.equal((tab[h, :37] >> (8 * b)) & 0xFF, e[:, 4 * h + b].to(torch.int64))
    dq = mx.mx8_dequantize(e8, e)
    blk = q.reshape(37, 12, 32)
    err = (dq.reshape(37, 12, 32) - blk).abs().amax(-1)
    amax = blk.abs().amax(-1)
    # e4m3: 3 mantissa bits (half an ulp = 2^-4 of the element's binade) — and the OCP scale 2^(floor(log2 amax) - 8) lets a block maximum in the top eighth of
    # its binade saturate at 448 (up to 2^-3 of itself)
    assert (err <= amax * 2.0 ** -3 + 1e-30).all()
    assert ((err <= amax * 2.0 ** -4 + 1e-30).float().mean() > 0.8)


def test_zero_scale_code_means_nothing_attends():
    """Keys past the end of K read scale code 0 (2^-127) and zero bytes through the buffer descriptor: their scores are exactly 0 before the kernel masks them."""
    z = mx.mx8_dequantize(torch.zeros((4, 128), dtype=torch.float8_e4m3fn), torch.zeros((4, 4), dtype=torch.uint8))
    assert torch.equal(z, torch.zeros(4, 128))


def test_what_the_mode_costs_depends_on_how_peaked_the_rows_are():
    """The numbers the docs quote: ~4e-2 on unit-variance operands, tenths on harsh channel magnitudes (peaked rows) — and the distance is a property of the
    quantised operands, the same whichever way the attention over them is evaluated."""
    for spread, lo, hi in ((0.0, 1e-2, 8e-2), (1.0, 5e-2, 0.8)):
        q, k, v = operands(7, 192, 640, 2, spread)
        exact = attention64(q, k, v, 2)
        qd = mx.mx8_dequantize(*mx.mx8_quantize(q))
        kd = mx.mx8_dequantize(*mx.mx8_quantize(k))
        quant = attention64(qd, kd, v, 2)
        d = rel(quant, exact)
        assert lo < d < hi, (spread, d)
        # rounding P to bf16 (what the kernel feeds P·V with) is second order next to it
        qh = qd.double().view(192, 2, 128).transpose(0, 1)
        kh = kd.double().view(640, 2, 128).transpose(0, 1)
        s = qh @ kh.transpose(1, 2) / 128 ** 0.5
        p = torch.exp(s - s.amax(-1, keepdim=True))
        pb = p.float().to(torch.bfloat16).double()
        vh = v.double().view(640, 2, 128).transpose(0, 1)
        o = ((pb @ vh) / p.sum(-1, keepdim=True)).transpose(0, 1).reshape(192, 256)
        assert rel(o, quant) < 3e-3


def test_bench_knows_the_flag():
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert '"--fp8-attn"' in src and 'SVI_ATTN_QK8' in src and "never the headline" in src


def test_python_switch_is_the_environment_switch_and_moves_the_graph_key():
    """svi_hip.fp8_attention() = SVI_ATTN_QK8 + a switch reload: off by default, and every flip moves the count captured step graphs are keyed on."""
    import os
    import svi_hip
    from svi_hip import _lib as L
    assert not svi_hip.fp8_attention_enabled()
    e0, g0 = L.switch_epoch(), None
    try:
        svi_hip.fp8_attention(True)
        assert os.environ.get("SVI_ATTN_QK8") == "1" and svi_hip.fp8_attention_enabled() and L.switch_epoch() == e0 + 1
    finally:
        svi_hip.fp8_attention(False)
    assert "SVI_ATTN_QK8" not in os.environ and not svi_hip.fp8_attention_enabled() and L.switch_epoch() == e0 + 2
