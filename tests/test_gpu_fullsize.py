"""-m gpu: the hot path at BASELINE.json's full sizes (C2: Wan2.1-1.3B widths, L = 21x30x52 = 32760 tokens), where the CPU
oracle takes hours.  Checked here: sampled rows against fp64 (the operators are row-separable: a sampled query row / output
row needs the whole K, V / whole weight but nothing of the other rows), size-independent properties (determinism, the CFG pair
and the context cache reproducing separate forwards bit for bit), and finiteness.  Inputs are generated on the GPU from seeds.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

L2, D, F, HEADS = 21 * 30 * 52, 1536, 8960, 12


@pytest.fixture(scope="module")
def hip():
    import svi_hip
    return svi_hip


def _rnd(seed, *shape, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def test_self_attention_c2_sampled_rows_vs_fp64(hip):
    """flash_fwd2 at L = 32760, 12 heads: 96 query rows spread over the sequence (first / last rows, block boundaries, the ragged
    last block) against fp64 softmax(q k^T / sqrt(128)) v over all 32760 keys.  Bound as for the small shapes: rel-L2 <= 6e-3."""
    q, k, v = _rnd(1, 1, L2, D), _rnd(2, 1, L2, D), _rnd(3, 1, L2, D)
    out = hip.flash_attention(q, k, v, HEADS)
    assert out.shape == (1, L2, D) and torch.isfinite(out.float()).all()
    rows = torch.tensor(sorted(set([0, 1, 63, 64, 255, 256, 257, 4095, 16383, 16384, L2 - 257, L2 - 256, L2 - 2, L2 - 1] +
                                   list(range(7, L2, L2 // 82)))), device="cuda")
    qs = q[0, rows].double().view(len(rows), HEADS, 128).transpose(0, 1)              # [h, r, d]
    kk = k[0].double().view(L2, HEADS, 128).permute(1, 2, 0)                           # [h, d, L]
    vv = v[0].double().view(L2, HEADS, 128).transpose(0, 1)                           # [h, L, d]
    p = torch.softmax(qs @ kk / 128 ** 0.5, dim=-1)
    want = (p @ vv).transpose(0, 1).reshape(len(rows), D)
    r = _rel(out[0, rows], want)
    assert r < 6e-3, r
    assert torch.equal(out, hip.flash_attention(q, k, v, HEADS))                       # deterministic


@pytest.mark.parametrize("name,M,N,K,epi", [("qkv", L2, D, D, "bias"), ("ffn1", L2, F, D, "gelu"), ("ffn2", L2, D, F, "gate_res")])
def test_gemm_c2_sampled_rows_vs_fp64(hip, name, M, N, K, epi):
    """The 256^2 LDS-DMA kernel at the C2 GEMM shapes (interior tiles, the ragged last row panel M = 127 x 256 + 248, N = 35
    column panels for ffn1): sampled output rows against fp64 with the epilogue's bf16 rounding points."""
    from svi_hip import _lib as L
    x, w, b = _rnd(10, M, K), _rnd(11, N, K, scale=K ** -0.5), _rnd(12, N)
    res, gate = _rnd(13, M, N), torch.randn(N, generator=torch.Generator(device="cuda").manual_seed(14), device="cuda")
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    code = {"bias": L.EPI_BIAS, "gelu": L.EPI_BIAS_GELU_TANH, "gate_res": L.EPI_BIAS_GATE_RES}[epi]
    L.check(L.lib().svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, code,
                                  gate.data_ptr() if epi == "gate_res" else None, res.data_ptr() if epi == "gate_res" else None, N,
                                  L.current_stream()))
    rows = torch.tensor([0, 1, 127, 128, 255, 256, 511, 12345, 32511, 32512, M - 249, M - 248, M - 2, M - 1], device="cuda")
    y = (x[rows].double() @ w.double().t() + b.double()).to(torch.bfloat16).double()         # nn.Linear output in bf16
    if epi == "gelu":
        y = torch.nn.functional.gelu(y, approximate="tanh")
    elif epi == "gate_res":
        y = res[rows].double() + (gate.double() * y).to(torch.bfloat16).double()
    r = _rel(out[rows], y)
    assert torch.isfinite(out.float()).all() and r < 4e-3, (name, r)


def _wan13b_two_blocks(hip, n_handles=1, layers=2):
    import synth
    c = dict(synth.WAN_1_3B)
    c["num_layers"] = layers
    g = torch.Generator(device="cuda").manual_seed(5)
    sd = {}
    for name, shape in synth.dit_param_shapes(**c).items():
        leaf = name.rsplit(".", 1)[-1]
        fan = 1
        for s_ in shape[1:]:
            fan *= s_
        if leaf == "modulation":
            t = torch.randn(shape, generator=g, device="cuda") / shape[-1] ** 0.5
        elif leaf == "weight" and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device="cuda")
        elif leaf == "weight":
            t = (torch.rand(shape, generator=g, device="cuda") * 2 - 1) / fan ** 0.5
        else:
            t = (torch.rand(shape, generator=g, device="cuda") * 2 - 1) * 0.05
        sd[name] = t.to(torch.bfloat16).contiguous()
    out = []
    for _ in range(n_handles):
        m = hip.WanDiT(eps=1e-6, num_heads=12, **c)
        m.bind(sd)
        out.append(m)
    return out


def test_dit_c2_geometry_pair_cache_determinism(hip):
    """Wan2.1-1.3B widths, 2 blocks, the full C2 latent [1,16,21,60,104]: two runs bit-equal; the CFG pair entry point and the
    context cache reproduce two separate uncached forwards bit for bit; outputs finite."""
    m = _wan13b_two_blocks(hip)[0]
    x = _rnd(20, 1, 16, 21, 60, 104)
    cp, cn = _rnd(21, 1, 512, 4096), _rnd(22, 1, 512, 4096)
    t = torch.tensor([991.7355])
    a, b = m.forward(x, t, cp).clone(), m.forward(x, t, cn).clone()
    assert torch.isfinite(a.float()).all() and not torch.equal(a, b)
    assert torch.equal(a, m.forward(x, t, cp))
    m.context_cache(True)
    try:
        for _ in range(2):
            pa, pb = m.forward_cfg_pair(x, t, cp, cn)
            assert torch.equal(pa, a) and torch.equal(pb, b)
    finally:
        m.context_cache(False)


@pytest.mark.parametrize("P", [2, 4])
def test_sequence_parallel_c2_geometry(hip, P):
    """The Ulysses schedule at C2 size: shards of 16380 / 8190 rows (ragged against every tile size), 6 / 3 heads per rank, the
    256^2 GEMM and the long-sequence attention kernel on shard-local buffers — same bits as the single-rank forward with the attention's
    key axis in one piece (SVI_FLASH_SPLIT=1).  By default a rank whose heads x q-blocks fill the chip's last round poorly (P = 4: 3 x 128
    = 384 workgroups on 256 compute units) cuts the key axis in two and merges the halves: the same softmax with another rounding
    sequence — equal to the single-rank forward to bf16 rounding (rel-L2 <= 4e-3 on the two-block output), not bit for bit."""
    from svi_hip import sequence_parallel as sp, _lib as L
    from gpu_util import report
    ms = _wan13b_two_blocks(hip, P + 1)
    x, ctx, t = _rnd(20, 1, 16, 21, 60, 104), _rnd(21, 1, 512, 4096), torch.tensor([991.7355])
    want = ms[-1].forward(x, t, ctx)
    got = sp.forward_local(ms[:P], x, t, ctx)
    same = bool(torch.equal(got, want))
    r = _rel(got.float().cpu(), want.float().cpu())
    report("sp_c2_geometry_default_vs_single_rank", P=P, identical=same, rel_l2=r)
    assert torch.isfinite(got.float()).all() and r < 4e-3, r
    if P == 2:
        assert same                      # 6 x 128 = 768 workgroups: three whole rounds, nothing to split
    L.set_switch("SVI_FLASH_SPLIT", 1)
    try:
        whole = sp.forward_local(ms[:P], x, t, ctx)
    finally:
        L.set_switch("SVI_FLASH_SPLIT", None)
    assert torch.equal(whole, want)


def test_dit_c2_geometry_vs_cpu_oracle(hip):
    """The whole forward at the full C2 geometry (Wan2.1-1.3B widths, L = 32760, one block) against the CPU oracle in fp32 on the
    same bf16-valued weights and inputs (about a minute of host time on the GPU box): the one place where every kernel's
    full-size code path (256^2 GEMMs, long-sequence attention, 3 KiB row kernels) meets the reference's arithmetic end to end.
    Bound: the whole-forward fp32 bound of the small cases, rel-L2 <= 2e-2."""
    from oracle import wan_dit_oracle as wdo
    import synth
    m = _wan13b_two_blocks(hip, layers=1)[0]
    x, ctx, t = _rnd(20, 1, 16, 21, 60, 104), _rnd(21, 1, 512, 4096), torch.tensor([991.7355])
    ctx[:, 64:] = 0                                                     # zero-padded prompt, as wan_prompter.py:107-108 leaves it
    got = m.forward(x, t, ctx).float().cpu()
    sd = {k: v.float().cpu() for k, v in m._params.items()}
    c = dict(synth.WAN_1_3B)
    cfg = wdo.DiTConfig(dim=c["dim"], in_dim=c["in_dim"], ffn_dim=c["ffn_dim"], out_dim=c["out_dim"], text_dim=c["text_dim"],
                        freq_dim=c["freq_dim"], patch_size=c["patch_size"], num_heads=12, num_layers=1, has_image_input=False)
    with torch.no_grad():
        want = wdo.dit_forward(sd, cfg, x.float().cpu(), t, ctx.float().cpu())
    r = _rel(got, want)
    from gpu_util import report
    report("dit_forward_c2_geometry_vs_oracle_fp32", rel_l2=r)
    assert got.shape == want.shape and r < 2e-2, r


@pytest.mark.parametrize("grid", [(2, 9, 7), (3, 16, 16)])
def test_multi_row_rmsnorm_rope_is_bit_identical(hip, grid):
    """At the DiT's widths (dim = 1536: three whole 16-byte chunks per lane) RMSNorm (+RoPE from the per-token table) and LayerNorm (+modulate /
    affine) run four rows per wave with the next row requested ahead and the gain / modulation vectors held in registers; SVI_RMS_ROWS=0
    selects the generic one-row-per-wave kernels.
    Same arithmetic, same bits: two blocks at 1.3B widths, the plain forward and the stacked CFG pair (token index modulo the sample
    length), token counts that are / are not multiples of the 16 rows a workgroup walks (126, 768)."""
    from svi_hip import _lib as L
    m = _wan13b_two_blocks(hip)[0]
    f, h, w = grid
    x, ctx, t = _rnd(40, 1, 16, f, 2 * h, 2 * w), _rnd(41, 1, 512, 4096), torch.tensor([500.0])
    a = m.forward(x, t, ctx)
    pa = m.forward_cfg_pair(x, t, ctx, -ctx)
    L.set_switch("SVI_RMS_ROWS", 0)
    try:
        b = m.forward(x, t, ctx)
        pb = m.forward_cfg_pair(x, t, ctx, -ctx)
    finally:
        L.set_switch("SVI_RMS_ROWS", None)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    assert torch.equal(pa[0], pb[0]) and torch.equal(pa[1], pb[1])


def test_dit_720p_geometry(hip):
    """81 frames at 1280x720 (the I2V-720P configuration's grid, 21x45x80 = 75600 tokens) at 1.3B widths, one block: finite,
    deterministic, and the two-rank sequence-parallel schedule reproduces it — bit for bit with the attention's key axis in one piece
    (SVI_FLASH_SPLIT=1), to bf16 rounding by default (the schedule's head groups launch 2 heads x 296 q-blocks = 592 workgroups on 256
    compute units at a time, which the launcher cuts into two key halves: 2.5 instead of 3 rounds)."""
    from svi_hip import sequence_parallel as sp, _lib as L
    ms = _wan13b_two_blocks(hip, 3, layers=1)
    x, ctx, t = _rnd(30, 1, 16, 21, 90, 160), _rnd(31, 1, 512, 4096), torch.tensor([500.0])
    a = ms[-1].forward(x, t, ctx)
    assert a.shape == (1, 16, 21, 90, 160) and torch.isfinite(a.float()).all()
    assert torch.equal(a, ms[-1].forward(x, t, ctx))
    b = sp.forward_local(ms[:2], x, t, ctx)
    assert _rel(b.float().cpu(), a.float().cpu()) < 4e-3
    L.set_switch("SVI_FLASH_SPLIT", 1)
    try:
        assert torch.equal(a, sp.forward_local(ms[:2], x, t, ctx))
    finally:
        L.set_switch("SVI_FLASH_SPLIT", None)


@pytest.mark.parametrize("grid", [(2, 9, 7), (5, 30, 52), (21, 30, 52)])
def test_fused_q_k_projection_is_bit_identical(hip, grid):
    """The self-attention q and k projections as ONE N = 2 dim launch over the two bound weight tensors (tile columns below dim multiply by Wq, the others by
    Wk; nothing is packed) against two launches (SVI_QK_FUSED=0): per element the same kernel and the same k order, so the same bits — on the 128^2 kernel (126
    tokens), on the 256-row tiles (7800 tokens) and at the full C2 size; the plain forward, the stacked / paired CFG forward, and a sequence-parallel shard
    (whose N = 1536 GEMMs run the 256 x 192 tile: 1536 = 8 x 192, the split falls on a tile boundary)."""
    from svi_hip import _lib as L, sequence_parallel as sp
    ms = _wan13b_two_blocks(hip, 3)
    f, h, w = grid
    x, ctx, t = _rnd(50, 1, 16, f, 2 * h, 2 * w), _rnd(51, 1, 512, 4096), torch.tensor([640.0])
    a = ms[0].forward(x, t, ctx)
    pa = ms[0].forward_cfg_pair(x, t, ctx, -ctx)
    sa = sp.forward_local(ms[1:3], x, t, ctx) if (f * h * w) % 2 == 0 else None
    L.set_switch("SVI_QK_FUSED", 0)
    try:
        b = ms[0].forward(x, t, ctx)
        pb = ms[0].forward_cfg_pair(x, t, ctx, -ctx)
        sb = sp.forward_local(ms[1:3], x, t, ctx) if sa is not None else None
    finally:
        L.set_switch("SVI_QK_FUSED", None)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    assert torch.equal(pa[0], pb[0]) and torch.equal(pa[1], pb[1])
    if sa is not None:
        assert torch.equal(sa, sb)


def test_block_and_unstacked_pair_at_75600_tokens_720p(hip):
    """A size ABOVE the headline: 81 frames at 1280x720 -> latent 21x90x160 -> 75600 tokens, Wan2.1-1.3B widths.
      * one DiT block (svi_dit_block_forward: 256^2 GEMMs with 296 row panels, the long-sequence attention over 75600 keys, the fused cross-attention
        on 75600 rows) against oracle.dit_block_rows — the oracle's own block statements on sampled rows (self-attention against the K / V of every
        token) with the bf16 rounding statement: rel-L2 <= 8e-3, as the C2-size block;
      * the CFG pair at this size: 2 x 75600 rows of the widest activation pass 2 GiB, so forward_pair takes its UNSTACKED fallback
        (csrc/svi_dit.hip forward_pair) — outputs bit-identical to two separate forwards, finite."""
    import numpy as np
    import synth
    from oracle import wan_dit_oracle as wdo
    grid = (21, 45, 80)
    f, h, w = grid
    L = f * h * w
    assert L == 75600
    c = dict(synth.WAN_1_3B, num_layers=1)
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(990, **c).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=12, **c)
    x = torch.from_numpy(synth.randn(991, 1, L, D))
    ctx = torch.from_numpy(synth.randn(992, 1, 512, D))
    ctx[:, 70:] = 0
    tm = torch.from_numpy(0.5 * synth.randn(993, 1, 6, D))
    got = m.block_forward(0, x.cuda(), ctx.cuda(), tm.cuda(), grid)
    assert torch.isfinite(got.float()).all() and torch.equal(got, m.block_forward(0, x.cuda(), ctx.cuda(), tm.cuda(), grid))
    rows = sorted(set([0, 1, 255, 256, 32759, 32760, 65535, 65536, L - 257, L - 2, L - 1] + list(range(11, L, L // 37))))
    sdb = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    cfg = wdo.DiTConfig(dim=D, in_dim=16, ffn_dim=F, out_dim=16, text_dim=4096, freq_dim=256, patch_size=(1, 2, 2), num_heads=12, num_layers=1, has_image_input=False)
    b16 = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    with torch.no_grad():
        want = wdo.dit_block_rows(sdb, "blocks.0.", b16(x), b16(ctx), b16(tm), wdo.rope_table_3d(128, grid), cfg, rows, rounding="bf16")
    r = _rel(got[0, rows].cpu(), want[0])
    from gpu_util import report
    report("dit_block_720p_75600_tokens", rel_l2_vs_oracle_bf16=r, rows=len(rows), tokens=L)
    assert r < 8e-3, r
    # the CFG pair at this size: unstacked fallback, same bits as two forwards
    lat = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=3, device="cpu", dtype=torch.float32).to("cuda", torch.bfloat16)
    cp = torch.from_numpy(synth.text_context(994, 512, 4096, 64)).to("cuda", torch.bfloat16)
    cn = torch.from_numpy(synth.text_context(995, 512, 4096, 32)).to("cuda", torch.bfloat16)
    ts = torch.tensor([700.0])
    m.context_cache(True)
    try:
        a, b = m.forward_cfg_pair(lat, ts, cp, cn)
        assert torch.isfinite(a.float()).all() and torch.isfinite(b.float()).all()
        assert torch.equal(a, m.forward(lat, ts, cp)) and torch.equal(b, m.forward(lat, ts, cn))
    finally:
        m.context_cache(False)
