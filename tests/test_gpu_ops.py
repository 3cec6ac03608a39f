"""-m gpu: every HIP operator, called through the C ABI, against the oracle's statement of the same op.

Tolerances (stated per test): inputs are bf16, accumulation fp32, outputs bf16.  Against the oracle's
"bf16 rounding" statement the only differences are accumulation order and a few fused roundings, so
rel-L2 <= 4e-3 (one bf16 ulp is 2^-8 = 3.9e-3 relative); against plain fp32 math, rel-L2 <= 1e-2.
"""
import math

import numpy as np
import pytest
import torch

import synth
from gpu_util import bf16r, dev, errs, host, report
from oracle import wan_dit_oracle as wdo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import svi_hip
    assert svi_hip._lib.lib().svi_device_count() >= 1
    return svi_hip


# ------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [
    (128, 128, 64), (256, 384, 128), (1, 1536, 256), (72, 128, 128), (257, 1280, 1280), (333, 200, 72),
    (1000, 1536, 1536), (512, 8960, 1536), (640, 1536, 8960), (130, 64, 1536),
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bias(hip, M, N, K):
    """A=I-style transposition traps are covered by asymmetric random operands and a non-square shape."""
    x = bf16r(torch.from_numpy(synth.randn(1, M, K)))
    w = bf16r(torch.from_numpy(synth.randn(2, N, K)) / math.sqrt(K))
    b = bf16r(torch.from_numpy(synth.randn(3, N)))
    want = (x.double() @ w.double().t() + b.double()).float()
    got = hip.linear(dev(x), dev(w), dev(b))
    r, mx, sc = errs(got, want)
    report("gemm_bias", M=M, N=N, K=K, rel_l2=r, max_abs=mx, scale=sc)
    assert got.shape == (M, N)
    assert r < 4e-3 and mx < 2 ** -7 * sc * 2, (r, mx, sc)


def test_gemm_identity_detects_transposes(hip):
    """W = shifted identity with asymmetric scaling: any swap of row/col or k-order shows up exactly."""
    K = N = 256
    M = 192
    x = bf16r(torch.from_numpy(synth.randn(5, M, K)))
    w = torch.zeros(N, K)
    for n in range(N):
        w[n, (n * 7 + 3) % K] = float(1 + (n % 5))
    got = host(hip.linear(dev(x), dev(w), None))
    want = bf16r(x @ w.t())
    assert torch.equal(got, want)


@pytest.mark.parametrize("epi", ["gelu_tanh", "gelu_erf", "silu", "gate_res", "res_nogate", "transposed"])
def test_gemm_epilogues(hip, epi):
    L = hip._lib
    M, N, K = 200, 264, 136
    x = bf16r(torch.from_numpy(synth.randn(11, M, K)))
    w = bf16r(torch.from_numpy(synth.randn(12, N, K)) / math.sqrt(K))
    b = bf16r(torch.from_numpy(synth.randn(13, N)))
    y = bf16r(x @ w.t() + b)
    if epi == "gelu_tanh":
        want = bf16r(wdo.gelu_tanh(y)); got = hip.linear(dev(x), dev(w), dev(b), epilogue=L.EPI_BIAS_GELU_TANH)
    elif epi == "gelu_erf":
        want = bf16r(0.5 * y * (1 + torch.erf(y / math.sqrt(2)))); got = hip.linear(dev(x), dev(w), dev(b), epilogue=L.EPI_BIAS_GELU_ERF)
    elif epi == "silu":
        want = bf16r(y * torch.sigmoid(y)); got = hip.linear(dev(x), dev(w), dev(b), epilogue=L.EPI_BIAS_SILU)
    elif epi in ("gate_res", "res_nogate"):
        res = bf16r(torch.from_numpy(synth.randn(14, M, N)))
        gate = bf16r(torch.from_numpy(synth.randn(15, N))) if epi == "gate_res" else None
        t = bf16r(gate * y) if gate is not None else y
        want = bf16r(res + t)
        got = hip.linear(dev(x), dev(w), dev(b), epilogue=L.EPI_BIAS_GATE_RES,
                         gate=None if gate is None else dev(gate, torch.float32), residual=dev(res))
    else:
        bm = b                      # bias runs along the rows of the transposed output
        want = bf16r((x @ w.t() + bm)).t()
        got = hip.linear(dev(x), dev(w), dev(b), transpose_out=True)[:, :M]
    r, mx, sc = errs(got, want)
    report("gemm_epilogue", epi=epi, rel_l2=r, max_abs=mx)
    assert r < 4e-3, (epi, r, mx)


@pytest.mark.parametrize("M,N,K,epi", [
    (8190, 1536, 1536, "bias"), (700, 1536, 1536, "gate_res"), (513, 8960, 256, "gelu_tanh"), (300, 200, 64, "bias"), (256, 192, 128, "gate_res"),
    (1030, 1000, 192, "gelu_tanh"), (1536, 1100, 128, "transposed"), (257, 392, 8960, "gate_res"), (777, 520, 128, "bias"), (2048, 512, 192, "gelu_tanh"),
    (2100, 8960, 256, "gelu_tanh"), (32760, 1536, 320, "gate_res"), (16380, 3072, 256, "bias")])      # more tiles than compute units
def test_gemm_tile_256x192_bit_identical(hip, M, N, K, epi):
    """The 256 x 192 tile kernel (what a sequence-parallel rank's shard GEMMs run on when 256-wide tiles fill the chip's rounds poorly)
    and the 256^2 tile in its two schedules (SVI_GEMM_KERNEL=259 / 260: four / two phases per K tile — wave rows one barrier apart, counted
    vmcnt, operands through buffer descriptors, rows past M / N read as zeros) against the 128^2 register-staged kernel on the same operands: per element the same K summation order and the same epilogue arithmetic, so the same
    bits — whole tiles, ragged row and column edges (N not a multiple of 192 or 8: the read-back's idle column chunks and the
    element-wise tail), one and many K tiles, every epilogue the DiT uses, the transposed-bias form — and against fp64 (rel-L2 <= 4e-3)."""
    L = hip._lib
    x = dev(synth.randn(31, M, K)); w = dev(synth.randn(32, N, K) / math.sqrt(K))
    b = dev(synth.randn(33, M if epi == "transposed" else N))
    gate = dev(synth.randn(34, N), torch.float32) if epi == "gate_res" else None
    res = dev(synth.randn(35, M, N)) if epi == "gate_res" else None
    kw = dict(epilogue={"bias": L.EPI_BIAS, "gate_res": L.EPI_BIAS_GATE_RES, "gelu_tanh": L.EPI_BIAS_GELU_TANH, "transposed": L.EPI_BIAS}[epi])
    if epi == "gate_res":
        kw.update(gate=gate, residual=res)
    outs = {}
    try:
        for kind in (128, 192, 259, 260):
            L.set_switch("SVI_GEMM_KERNEL", kind)
            if epi == "transposed":      # C^T = W X^T with the bias along the rows of the output (the DiT's V^T projection)
                out = torch.zeros((N, (M + 7) // 8 * 8), dtype=torch.bfloat16, device="cuda")
                L.check(L.lib().svi_gemm_bf16(w.data_ptr(), K, x.data_ptr(), K, out.data_ptr(), out.shape[1], N, M, K, dev(synth.randn(33, N)).data_ptr(), 1,
                                              L.EPI_BIAS, None, None, 0, L.current_stream()))
                outs[kind] = out[:, :M].clone()
            else:
                outs[kind] = hip.linear(x, w, b, **kw)
    finally:
        L.set_switch("SVI_GEMM_KERNEL", None)
    assert torch.equal(outs[192], outs[128]) and torch.equal(outs[259], outs[128]) and torch.equal(outs[260], outs[128])
    if epi == "bias":
        want = (x.double().cpu() @ w.double().cpu().t() + b.double().cpu()).float()
        r, mx, _ = errs(outs[192], want)
        report("gemm_tile_256x192", M=M, N=N, K=K, rel_l2=r, max_abs=mx)
        assert r < 4e-3, r


@pytest.mark.parametrize("M,N,K,epi", [
    (2560, 1536, 8960, "gate_res"), (2560, 1536, 1536, "bias"), (1283, 1536, 1536, "gate_res"), (1280, 1000, 256, "gelu_tanh"), (640, 3072, 320, "bias"),
    (1536, 2560, 1536, "transposed"), (300, 200, 448, "bias")])
def test_gemm_128_tile_deep_prefetch_bit_identical(hip, M, N, K, epi):
    """The 128^2 kernel with four K tiles of loads in flight, on four waves and on eight (64 x 32 per wave, two per SIMD: what a launch of at most one
    workgroup per CU takes — the C1-size step's projections, 240 tiles of M = 2560 stacked rows) against the same kernel with one (SVI_GEMM_PF = 4 / 8 / 1,
    all forced onto the 128^2 tile): same k order per element, so the same bits — K of 4 (no steady-state iteration), 5, 7, 24 and 140 tiles, ragged row and column edges, every epilogue of the block; and against fp64."""
    L = hip._lib
    x = dev(synth.randn(41, M, K)); w = dev(synth.randn(42, N, K) / math.sqrt(K))
    b = dev(synth.randn(43, N))
    gate = dev(synth.randn(44, N), torch.float32) if epi == "gate_res" else None
    res = dev(synth.randn(45, M, N)) if epi == "gate_res" else None
    kw = dict(epilogue={"bias": L.EPI_BIAS, "gate_res": L.EPI_BIAS_GATE_RES, "gelu_tanh": L.EPI_BIAS_GELU_TANH, "transposed": L.EPI_BIAS}[epi])
    if epi == "gate_res":
        kw.update(gate=gate, residual=res)
    outs = {}
    try:
        L.set_switch("SVI_GEMM_KERNEL", 128)
        for pf in (1, 4, 8):
            L.set_switch("SVI_GEMM_PF", pf)
            outs[pf] = hip.linear(x, w, b, transpose_out=True)[:, :M].clone() if epi == "transposed" else hip.linear(x, w, b, **kw)
    finally:
        L.set_switch("SVI_GEMM_KERNEL", None)
        L.set_switch("SVI_GEMM_PF", None)
    assert torch.equal(outs[4], outs[1]) and torch.equal(outs[8], outs[1])
    if epi == "bias":
        want = (x.double().cpu() @ w.double().cpu().t() + b.double().cpu()).float()
        r, mx, _ = errs(outs[8], want)
        report("gemm_128_deep_prefetch", M=M, N=N, K=K, rel_l2=r, max_abs=mx)
        assert r < 4e-3, r


def test_gemm_inplace_residual(hip):
    """x += gate*(h W^T + b) with the residual aliasing the output, as the block does (dit:369,373)."""
    L = hip._lib
    M, N, K = 300, 256, 128
    h = dev(synth.randn(21, M, K)); w = dev(synth.randn(22, N, K) / math.sqrt(K)); b = dev(synth.randn(23, N))
    gate = dev(synth.randn(24, N), torch.float32)
    x = dev(synth.randn(25, M, N))
    ref = hip.linear(h, w, b, epilogue=L.EPI_BIAS_GATE_RES, gate=gate, residual=x.clone())
    xin = x.clone()
    L.check(L.lib().svi_gemm_bf16(h.data_ptr(), K, w.data_ptr(), K, xin.data_ptr(), N, M, N, K, b.data_ptr(), 0,
                                  L.EPI_BIAS_GATE_RES, gate.data_ptr(), xin.data_ptr(), N, L.current_stream()))
    assert torch.equal(xin, ref)


# ------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("rows,dim", [(1, 128), (7, 1536), (130, 1280), (33, 5120), (5, 8192)])
@pytest.mark.parametrize("mode", ["plain", "affine", "mod"])
def test_layernorm_modulate(hip, rows, dim, mode):
    x = bf16r(torch.from_numpy(3 * synth.randn(31, rows, dim) + 0.5))
    kw, want = {}, bf16r(wdo.layer_norm(x, 1e-6))
    if mode == "affine":
        w = bf16r(torch.from_numpy(1 + 0.1 * synth.randn(32, dim))); b = bf16r(torch.from_numpy(0.1 * synth.randn(33, dim)))
        want = bf16r(wdo.layer_norm(x, 1e-6, w, b)); kw = dict(weight=dev(w), bias=dev(b))
    elif mode == "mod":
        sh = bf16r(torch.from_numpy(0.3 * synth.randn(34, dim))); sc = bf16r(torch.from_numpy(0.3 * synth.randn(35, dim)))
        want = wdo.modulated_norm(x[None], sh[None, None], sc[None, None], 1e-6, bf16r)[0]
        kw = dict(shift=dev(sh), scale=dev(sc))
    got = hip.layernorm_modulate(dev(x), 1e-6, **kw)
    r, mx, sc_ = errs(got, want)
    report("layernorm", rows=rows, dim=dim, mode=mode, rel_l2=r, max_abs=mx)
    assert r < 2e-3, (r, mx)


@pytest.mark.parametrize("grid,heads", [((1, 1, 1), 1), ((3, 4, 6), 1), ((2, 5, 7), 2), ((2, 3, 5), 12), ((21, 2, 3), 2)])
def test_rmsnorm_rope(hip, grid, heads):
    f, h, w = grid
    L, dim = f * h * w, heads * 128
    x = bf16r(torch.from_numpy(2 * synth.randn(41, L, dim)))
    wt = bf16r(torch.from_numpy(1 + 0.1 * synth.randn(42, dim)))
    n = wdo.rms_norm_full(x[None], wt, 1e-6, bf16r)
    want_norm = n[0]
    want_rope = bf16r(wdo.apply_rope(n, wdo.rope_table_3d(128, grid), heads))[0]
    got_norm = hip.rmsnorm_rope_(dev(x), dev(wt), 1e-6)
    got_rope = hip.rmsnorm_rope_(dev(x), dev(wt), 1e-6, grid=grid, num_heads=heads)
    r0, m0, _ = errs(got_norm, want_norm)
    r1, m1, _ = errs(got_rope, want_rope)
    report("rmsnorm_rope", grid=list(grid), heads=heads, rel_l2_norm=r0, rel_l2_rope=r1, max_abs_rope=m1)
    assert r0 < 2e-3 and r1 < 2e-3, (r0, r1)


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (1000, 1536, 256), (8190, 1536, 1536), (2049, 5120, 128), (130, 192, 72)])
def test_linear_row_stats_is_one_tree_in_every_tile_kernel(hip, M, N, K):
    """svi_linear_row_stats: the q projection whose epilogue also leaves RMSNorm's statistic.  (a) C is bit for bit svi_gemm_bf16's; (b) the per-64-column
    sums of squares and rs are BITWISE the same whichever tile kernel ran (128^2, 256 x 192, 256^2 in both schedules: one summation tree), and whether
    the rows came as one launch or as shards (a sequence-parallel rank's rows); (c) against fp64 on the rounded outputs: rel 1e-6."""
    L = hip._lib
    x = dev(synth.randn(61, M, K)); w = dev(synth.randn(62, N, K) / math.sqrt(K)); b = dev(synth.randn(63, N))
    outs = {}
    try:
        for kind in (128, 192, 259, 260):
            L.set_switch("SVI_GEMM_KERNEL", kind)
            outs[kind] = hip.ops.linear_row_stats(x, w, b, eps=1e-6)
    finally:
        L.set_switch("SVI_GEMM_KERNEL", None)
    y, rs, ss = outs[128]
    assert torch.equal(y, hip.linear(x, w, b))
    for kind in (192, 259, 260):
        assert torch.equal(outs[kind][0], y) and torch.equal(outs[kind][2], ss) and torch.equal(outs[kind][1], rs), kind
    h = M // 2 + 3
    top, bot = hip.ops.linear_row_stats(x[:h].contiguous(), w, b), hip.ops.linear_row_stats(x[h:].contiguous(), w, b)
    assert torch.equal(torch.cat([top[1], bot[1]]), rs)
    y64 = y.double().cpu()
    want_ss = (y64 * y64).reshape(M, N // 64, 64).sum(-1).t()
    want_rs = 1.0 / torch.sqrt((y64 * y64).mean(-1) + 1e-6)
    r_ss, r_rs = errs(ss, want_ss.float())[0], errs(rs, want_rs.float())[0]
    report("linear_row_stats", M=M, N=N, K=K, rel_sumsq=r_ss, rel_rs=r_rs)
    assert r_ss < 1e-6 and r_rs < 1e-6, (r_ss, r_rs)


@pytest.mark.parametrize("Lq,Lk,heads,tail", [(128, 64, 1, None), (1000, 512, 2, (65, 448)), (333, 257, 3, None), (4100, 512, 12, (33, 480)), (77, 512, 2, None), (2050, 33, 12, (33, 1)),
                                                  (700, 512, 2, (200, 313)), (300, 96, 1, None), (513, 128, 2, (128, 1)), (260, 130, 1, (129, 2))])
def test_cross_attention_normalises_q_as_it_reads_it(hip, Lq, Lk, heads, tail):
    """svi_cross_attention_fwd (up to 128 walked keys: flash_cross_resident_kernel, K / V^T resident in LDS; more: flash_fwd_kernel<1, STAGED>; with a device-side
    key_tail both are launched and gate themselves): (a) without the normalisation it gives the plain short-key kernel's result on the same operands; (b) with it — q raw, rs from svi_linear_row_stats' arithmetic — it equals normalising q first
    with svi_rmsnorm_rope's rounding points and then attending: bit for bit when the two statistics agree, and they are compared here (the standalone
    kernel sums a row's squares in another order: the fp32 statistic may differ in the last bit, which moves a bf16 rounding of q on rare elements;
    measured rate reported, bound 2e-3 of the elements, output rel-L2 <= 1e-3); (c) against fp64 attention on the normalised q: rel-L2 <= 6e-3."""
    D = heads * 128
    qraw = dev(2.0 * synth.randn(71, Lq, D))
    gain = dev(1 + 0.1 * synth.randn(72, D))
    k = dev(synth.randn(73, Lk, D))
    v = bf16r(torch.from_numpy(synth.randn(74, Lk, D)))
    ldvt = (Lk + 7) // 8 * 8
    vt = torch.zeros((D, ldvt), dtype=torch.bfloat16, device="cuda")
    vt[:, :Lk] = dev(v).t()
    SC = 0.12751743          # softmax_scale * log2(e) for head_dim 128 (SVI_QK_SCALE_LOG2E)
    kt = None
    if tail is not None:     # keys n-1 .. Lk-1 identical: make them so
        n, m = tail
        k[n - 1:] = k[n - 1]
        vt[:, n - 1:Lk] = vt[:, n - 1:n]
        kt = torch.tensor([n, m], dtype=torch.int32, device="cuda")
        assert n - 1 + m == Lk
    # the two-kernel path: RMSNorm in place (out_scale folded in by the library's own call) then the short-key attention
    lib = hip._lib
    qn = qraw.clone()
    lib.check(lib.lib().svi_rmsnorm_rope(qn.data_ptr(), D, Lq, D, gain.data_ptr(), 1e-6, 0, 0, 0, 0, 0, lib.current_stream()))
    qn_scaled = (qn.float() * SC).to(torch.bfloat16)          # the DiT folds SC into the same final rounding; here it is one more rounding, so (b) compares against the fused kernel fed the same way
    plain = hip.ops.cross_attention(qn_scaled, k, vt, heads, s_kv=Lk, key_tail=kt)
    ref_kernel = torch.empty_like(plain)
    lib.check(lib.lib().svi_attention_vt_fwd(qn_scaled.data_ptr(), D, k.data_ptr(), D, vt.data_ptr(), ldvt, ref_kernel.data_ptr(), D, Lq, Lk if tail is None else tail[0],
                                             heads, 1, lib.current_stream()))
    if tail is None:                                          # (a)
        if Lk <= 64 or Lk > 128:      # one key tile, or the streaming kernel itself: the very same arithmetic
            assert torch.equal(plain, ref_kernel)
        else:                         # 65 .. 128 keys: the resident kernel's softmax is exact in one sweep, the streaming kernel's online (one rescale): same to rounding
            assert errs(plain, ref_kernel)[0] < 2e-3
    # (b): statistic in the projection-epilogue's tree — feed the raw q through an identity "projection" to get rs
    eye = torch.eye(D, dtype=torch.bfloat16, device="cuda")
    y, rs, _ = hip.ops.linear_row_stats(qraw, eye, None, eps=1e-6)
    assert torch.equal(y, qraw)
    rs_ref = 1.0 / torch.sqrt((qraw.double() ** 2).mean(-1) + 1e-6)
    assert float(((rs.double() - rs_ref).abs() / rs_ref).max()) < 1e-6
    fused1 = hip.ops.cross_attention(qraw, k, vt, heads, s_kv=Lk, q_rs=rs, q_gain=gain, q_out_scale=1.0, key_tail=kt)      # q' = norm(q), no extra scale
    two = hip.ops.cross_attention(qn, k, vt, heads, s_kv=Lk, key_tail=kt)
    frac = float((fused1 != two).float().mean())
    r_b = errs(fused1, two)[0]
    # (c)
    vv = v.double().clone()
    if tail is not None:
        vv[tail[0] - 1:] = vv[tail[0] - 1]
    # fp64 softmax(q'.k^T) v over ALL Lk keys, q' carrying the scale in base-2 units (the kernel's convention): p = 2^(q'.k - max)
    qh = qn.double().cpu().reshape(Lq, heads, 128).transpose(0, 1)
    kh = k.double().cpu().reshape(Lk, heads, 128).transpose(0, 1)
    vh = vv.reshape(Lk, heads, 128).transpose(0, 1)
    sc = qh @ kh.transpose(1, 2)
    p = torch.exp2(sc - sc.max(-1, keepdim=True).values)
    want = ((p / p.sum(-1, keepdim=True)) @ vh).transpose(0, 1).reshape(Lq, D).float()
    r_c = errs(two, want)[0]
    r_cf = errs(fused1, want)[0]
    report("cross_attention_fused", Lq=Lq, Lk=Lk, heads=heads, tail=list(tail) if tail else None, frac_elements_differ=frac, rel_l2_fused_vs_two_kernel=r_b,
           rel_l2_two_kernel_vs_fp64=r_c, rel_l2_fused_vs_fp64=r_cf)
    assert frac < 2e-3 and r_b < 1e-3, (frac, r_b)
    assert r_c < 6e-3 and r_cf < 6e-3, (r_c, r_cf)
    # (d) the DiT's call: q_out_scale = softmax_scale * log2(e), folded into q' before its last rounding: softmax2(SC q'.k^T) v against fp64 on the normalised q.
    fused_sc = hip.ops.cross_attention(qraw, k, vt, heads, s_kv=Lk, q_rs=rs, q_gain=gain, q_out_scale=SC, key_tail=kt)
    sc2 = sc * SC
    p2 = torch.exp2(sc2 - sc2.max(-1, keepdim=True).values)
    want2 = ((p2 / p2.sum(-1, keepdim=True)) @ vh).transpose(0, 1).reshape(Lq, D).float()
    r_d = errs(fused_sc, want2)[0]
    r_d2 = errs(fused_sc, plain)[0]          # ... and against the two-kernel path fed q' * SC rounded once more
    report("cross_attention_fused_scaled", Lq=Lq, Lk=Lk, heads=heads, tail=list(tail) if tail else None, rel_l2_vs_fp64=r_d, rel_l2_vs_two_kernel=r_d2)
    assert r_d < 6e-3 and r_d2 < 6e-3, (r_d, r_d2)


def test_rmsnorm_strided_view(hip):
    """The block normalises q and k in place inside the [L, 2D] q|k buffer (ld = 2D)."""
    L_, D = 50, 256
    qk = dev(synth.randn(43, L_, 2 * D))
    wt = dev(1 + 0.1 * synth.randn(44, D))
    ref = hip.rmsnorm_rope_(qk[:, D:].contiguous(), wt, 1e-6)
    lib = hip._lib
    before_q = qk[:, :D].clone()
    lib.check(lib.lib().svi_rmsnorm_rope(qk.data_ptr() + D * 2, 2 * D, L_, D, wt.data_ptr(), 1e-6, 0, 0, 0, 0, 0,
                                         lib.current_stream()))
    assert torch.equal(qk[:, D:], ref) and torch.equal(qk[:, :D], before_q)


# ------------------------------------------------------------------------------------------ attention
ATT_SHAPES = [(64, 64, 1), (128, 512, 2), (100, 77, 1), (1280, 1280, 2), (333, 257, 3), (72, 72, 12), (1, 1, 1),
              (4096, 4096, 1), (130, 1000, 2)]


@pytest.mark.parametrize("Lq,Lk,heads", ATT_SHAPES)
def test_flash_attention(hip, Lq, Lk, heads):
    """vs softmax(QK^T/sqrt(128))V in fp64 on the same bf16 inputs.  P is rounded to bf16 before PV inside the
    kernel (as every flash-attention does), so the bound is a little over one bf16 ulp: rel-L2 <= 6e-3."""
    D = heads * 128
    q = bf16r(torch.from_numpy(synth.randn(51, 1, Lq, D)))
    k = bf16r(torch.from_numpy(synth.randn(52, 1, Lk, D)))
    v = bf16r(torch.from_numpy(synth.randn(53, 1, Lk, D)))
    want = wdo.attention(q.double(), k.double(), v.double(), heads).float()
    got = hip.flash_attention(dev(q), dev(k), dev(v), heads)
    r, mx, sc = errs(got, want)
    report("flash_attention", Lq=Lq, Lk=Lk, heads=heads, rel_l2=r, max_abs=mx, scale=sc)
    assert r < 6e-3, (r, mx)


@pytest.mark.parametrize("Lq,Lk,heads", [(64, 64, 1), (100, 100, 1), (192, 192, 2), (300, 1000, 1), (1000, 130, 2), (2304, 2304, 1), (4096, 4096, 1)])
def test_flash_attention_16x16x32_pass_matches_the_32x32x16_pass(hip, Lq, Lk, heads):
    """Round 6: the optimistic pass of the long-sequence attention runs on v_mfma_f32_16x16x32_bf16 (flash_fwd3_kernel; SVI_FLASH_M16=0 is the rounds 2-6 kernel on
    32x32x16).  Forced onto short and ragged key axes (one real tile, masked last tiles, Lq != Lk: every prologue / drain path of the tile loop): against fp64 within
    the attention tolerance, and against the 32x32x16 pass far inside it (same softmax, same bf16 rounding points; only the matrix instruction's summation order
    differs)."""
    L = hip._lib
    D = heads * 128
    q = dev(synth.randn(71, 1, Lq, D)); k = dev(synth.randn(72, 1, Lk, D)); v = dev(synth.randn(73, 1, Lk, D))
    want = wdo.attention(q.double().cpu(), k.double().cpu(), v.double().cpu(), heads).float()
    outs = {}
    try:
        L.set_switch("SVI_FLASH_KERNEL", 2)
        for m16 in (1, 0):
            L.set_switch("SVI_FLASH_M16", m16)
            outs[m16] = hip.flash_attention(q, k, v, heads).float().cpu()
    finally:
        L.set_switch("SVI_FLASH_M16", None)
        L.set_switch("SVI_FLASH_KERNEL", None)
    r16, mx16, _ = errs(outs[1], want)
    r32, _, _ = errs(outs[0], want)
    rab, _, _ = errs(outs[1], outs[0])
    report("flash_attention_m16", Lq=Lq, Lk=Lk, heads=heads, rel_l2=r16, rel_l2_32x32=r32, rel_l2_between=rab, max_abs=mx16)
    assert r16 < 6e-3 and r32 < 6e-3 and rab < 3e-3, (r16, r32, rab)


def test_flash_attention_online_softmax_rescale(hip):
    """Force the running max to jump late in the key axis (a spiked key in the last tile) and early (first tile):
    exercises the O/l rescale branch that bounded random data barely touches."""
    Lq, Lk, heads = 96, 448, 1
    q = bf16r(torch.from_numpy(synth.randn(61, 1, Lq, 128)))
    k = bf16r(torch.from_numpy(synth.randn(62, 1, Lk, 128)))
    v = bf16r(torch.from_numpy(synth.randn(63, 1, Lk, 128)))
    k[0, 430] = q[0, 17] * 4.0          # row 17's max jumps by ~45 logits at the last tile
    k[0, 3] = q[0, 40] * 4.0            # row 40 peaks in the first tile, everything later is rescaled against it
    want = wdo.attention(q.double(), k.double(), v.double(), heads).float()
    got = hip.flash_attention(dev(q), dev(k), dev(v), heads)
    r, mx, _ = errs(got, want)
    report("flash_attention_spike", rel_l2=r, max_abs=mx)
    assert r < 6e-3 and mx < 0.05, (r, mx)


@pytest.mark.parametrize("pattern", ["spikes", "ramp", "late_giant"])
def test_flash_attention_long_sequence_rescale_paths(hip, pattern):
    """The long-sequence kernel (Lk >= 2048) moves its reference maximum only when a row outgrows it by 2^8 and does so outside
    its steady-state loop: drive that path hard.  spikes: single keys 30-60 logits above the rest at tile boundaries and inside
    tiles, early and late; ramp: key norms grow along the sequence so every few tiles some row outgrows its reference;
    late_giant: the very last key dominates every row (everything accumulated before is rescaled to ~0)."""
    Lq, Lk, heads = 320, 4200, 2
    D = heads * 128
    q = bf16r(torch.from_numpy(synth.randn(91, 1, Lq, D)))
    k = bf16r(torch.from_numpy(synth.randn(92, 1, Lk, D)))
    v = bf16r(torch.from_numpy(synth.randn(93, 1, Lk, D)))
    if pattern == "spikes":
        for key, row, gain in [(0, 5, 3.0), (63, 70, 4.0), (64, 71, 4.0), (2047, 130, 3.5), (2048, 200, 5.0), (4199, 319, 4.0), (4100, 5, 5.0)]:
            k[0, key] = q[0, row] * gain
    elif pattern == "ramp":
        k = bf16r(k * torch.linspace(0.2, 3.0, Lk).view(1, Lk, 1))
    else:
        k[0, Lk - 1] = bf16r(q[0].mean(dim=0) * 0 + 1.0) * 2.0
        q = bf16r(q + 1.5)                                  # every row has a large positive projection on the last key
    want = wdo.attention(q.double(), k.double(), v.double(), heads).float()
    got = hip.flash_attention(dev(q), dev(k), dev(v), heads)
    r, mx, _ = errs(got, want)
    report("flash_attention_long_rescale", pattern=pattern, rel_l2=r, max_abs=mx)
    assert torch.isfinite(got.float()).all() and r < 6e-3, (pattern, r, mx)


@pytest.mark.parametrize("pieces,pattern", [(2, "plain"), (3, "plain"), (2, "ragged"), (2, "late_giant"), (2, "ramp")])
def test_flash_attention_split_key_axis(hip, pieces, pattern):
    """The long-sequence kernel with its key axis cut into pieces that run as separate workgroups (what a sequence-parallel rank's 3 heads x
    128 q-blocks need to fill 256 compute units evenly) and a merge of the pieces' unnormalised O, reference maxima and row sums: the
    same softmax(QK^T/sqrt(128))V (fp64 on the same bf16 inputs, rel-L2 <= 6e-3), including a key count that is not a multiple of the
    tile or of the piece length, q rows that end inside a workgroup, pieces whose maxima differ by hundreds of logits (the last key
    dominating every row: the earlier pieces merge to ~0), a norm ramp along the keys (later pieces outweigh earlier ones; the optimistic
    pass's flagged workgroups are recomputed per piece), and agreement with the unsplit kernel (SVI_FLASH_SPLIT=1) to bf16 rounding."""
    from svi_hip import _lib as L
    Lq, Lk, heads = (300, 8192 * pieces + 77, 2) if pattern == "ragged" else (320, 8192 * pieces, 2)
    D = heads * 128
    q = bf16r(torch.from_numpy(synth.randn(191, 1, Lq, D)))
    k = bf16r(torch.from_numpy(synth.randn(192, 1, Lk, D)))
    v = bf16r(torch.from_numpy(synth.randn(193, 1, Lk, D)))
    if pattern == "ramp":
        k = bf16r(k * torch.linspace(0.2, 3.0, Lk).view(1, Lk, 1))
    elif pattern == "late_giant":
        k[0, Lk - 1] = 2.0
        q = bf16r(q + 1.5)
    want = wdo.attention(q.double(), k.double(), v.double(), heads).float()
    L.set_switch("SVI_FLASH_SPLIT", pieces)
    try:
        got = hip.flash_attention(dev(q), dev(k), dev(v), heads)
        L.set_switch("SVI_FLASH_SPLIT", 1)
        whole = hip.flash_attention(dev(q), dev(k), dev(v), heads)
    finally:
        L.set_switch("SVI_FLASH_SPLIT", None)
    r, mx, _ = errs(got, want)
    rw, mw, _ = errs(got, whole.float().cpu())
    report("flash_attention_split", pieces=pieces, pattern=pattern, rel_l2=r, max_abs=mx, vs_unsplit_rel=rw, vs_unsplit_max=mw)
    assert torch.isfinite(got.float()).all() and r < 6e-3 and rw < 6e-3, (pieces, pattern, r, rw)


def test_flash_attention_cuts_only_the_last_round(hip):
    """The launcher's own decision: 100 q-blocks x 3 heads = 300 work items on 256 compute units leave 44 items for a second round; those
    (q-blocks 56..99 of the last head) are cut into 2 key pieces (8192 keys: two pieces of >= 4096) and merged, the 256 items of the first round run whole.
    Against the single-piece kernel (SVI_FLASH_SPLIT=1, itself checked against fp64 above): rows of uncut items carry the same bits, rows
    of cut items agree to bf16 rounding and are not all identical (the cut path really ran)."""
    from svi_hip import _lib as L
    Lq, Lk, heads = 25600, 8192, 3
    q = dev(synth.randn(195, 1, Lq, heads * 128)); k = dev(synth.randn(196, 1, Lk, heads * 128)); v = dev(synth.randn(197, 1, Lk, heads * 128))
    got = hip.flash_attention(q, k, v, heads)
    L.set_switch("SVI_FLASH_SPLIT", 1)
    try:
        whole = hip.flash_attention(q, k, v, heads)
    finally:
        L.set_switch("SVI_FLASH_SPLIT", None)
    g, w = got[0].float(), whole[0].float()
    cut_rows = slice(56 * 256, Lq)
    assert torch.equal(g[:, :256], w[:, :256]) and torch.equal(g[: 56 * 256, 256:], w[: 56 * 256, 256:])          # heads 0, 1 and the uncut q-blocks of head 2
    cg, cw = g[cut_rows, 256:], w[cut_rows, 256:]
    r = float((cg - cw).norm() / cw.norm())
    report("flash_attention_last_round_cut", rel_l2=r, identical=bool(torch.equal(cg, cw)))
    assert r < 6e-3 and not torch.equal(cg, cw), r


def test_flash_attention_batch_and_linearity_in_v(hip):
    """Size-independent property: attention is linear in V;  attn(q,k,a*v1+v2) == a*attn(q,k,v1)+attn(q,k,v2)."""
    q = dev(synth.randn(71, 2, 200, 256)); k = dev(synth.randn(72, 2, 300, 256))
    v1 = dev(synth.randn(73, 2, 300, 256)); v2 = dev(synth.randn(74, 2, 300, 256))
    o1 = hip.flash_attention(q, k, v1, 2).float(); o2 = hip.flash_attention(q, k, v2, 2).float()
    o12 = hip.flash_attention(q, k, (2 * v1.float() + v2.float()).to(torch.bfloat16), 2).float()
    r, mx, _ = errs(o12, 2 * o1 + o2)
    assert r < 1e-2, r


# ------------------------------------------------------------------------------------------ cfg step
def test_cfg_step_bit_exact(hip):
    n = 16 * 5 * 32 * 32 + 3
    lat = bf16r(torch.from_numpy(synth.randn(81, n))); c = bf16r(torch.from_numpy(synth.randn(82, n))); u = bf16r(torch.from_numpy(synth.randn(83, n)))
    s, ds = 5.0, -0.021739
    want = bf16r(lat + bf16r(bf16r(u + bf16r(s * bf16r(c - u))) * ds))
    got = host(hip.cfg_step_(dev(lat), dev(c), dev(u), s, ds))
    assert torch.equal(got, want)
    want1 = bf16r(lat + bf16r(c * ds))
    got1 = host(hip.cfg_step_(dev(lat), dev(c), None, 1.0, ds))
    assert torch.equal(got1, want1)


# ------------------------------------------------------------------------------------------ LoRA merge
@pytest.mark.parametrize("out_f,in_f,r,alpha", [(1536, 1536, 128, 1.0), (8960, 1536, 64, 0.7), (200, 264, 32, 2.0)])
def test_lora_merge_matches_reference_rounding(hip, out_f, in_f, r, alpha):
    """models/lora.py:246-262 in the model dtype: bf16(W + bf16(alpha * bf16(up @ down))).  The dot products are formed in fp64
    here; where fp32 accumulation order decides a bf16 rounding tie the result may differ by one ulp of the product term."""
    w = bf16r(torch.from_numpy(synth.randn(81, out_f, in_f)) * 0.05)
    up = bf16r(torch.from_numpy(synth.randn(82, out_f, r)) * 0.1)
    down = bf16r(torch.from_numpy(synth.randn(83, r, in_f)) * 0.1)
    prod = (up.double() @ down.double()).to(torch.bfloat16)
    want = (w.to(torch.bfloat16) + (alpha * prod)).float()            # torch: bf16 * python float -> bf16, bf16 + bf16 -> bf16
    wd = dev(w)
    ptr = wd.data_ptr()
    got = hip.lora.merge_lora_(wd, up, down, alpha)
    assert got.data_ptr() == ptr                                      # in place: a bound WanDiT keeps seeing the tensor
    r_, mx, _ = errs(got, want)
    frac = float((host(got) != want).float().mean())
    report("lora_merge", out_f=out_f, in_f=in_f, rank=r, rel_l2=r_, differing=frac)
    assert r_ < 1e-3 and frac < 0.02, (r_, frac)
    with pytest.raises(ValueError):
        hip.lora.merge_lora_(wd, up[:, :-8], down, alpha)
