"""-m gpu: the opt-in quantised-QK^T attention (SVI_ATTN_QK8).  The reference computes attention in bf16; its dispatch
(models/wan_video_dit.py:116-147) accepts a quantised-QK^T backend (SageAttention) as interchangeable, and that is the only sense in which this
mode has a reference counterpart: there are no reference numbers to pin it to.  The checker is therefore the MX format restated in
oracle/mx8_oracle.py plus fp64 attention, and the statements made are:

  kernel         with Q and K quantised to MX e4m3 (one E8M0 scale per 32 channels — the MLP's quantiser, bit-exact against the oracle in
                 test_gpu_mx8.py) the kernel equals softmax(q^ k^T / sqrt d) v over the DEQUANTISED operands to the bf16 kernel's own bound
                 (rel-L2 <= 6e-3): the fp8 MFMA and its block scales multiply exactly what the oracle says they hold
  second pass    adversarial operands (a late giant key) flag every workgroup; the complete fp8 kernel recomputes them: same bound
  distance       what the quantisation costs is REPORTED per call (it depends on how peaked the rows are: 4e-2 on unit-variance operands, 0.13-0.22 on the
                 deliberately harsh channel magnitudes used here — e4m3 keeps 3 mantissa bits) and shown to be ALL of the distance from the bf16 kernel;
                 it is BOUNDED where the mode is meant to be used: the 30-layer forward at 7800 tokens (dit_depth.npz, made by the reference) stays
                 <= 5e-2 of the bf16-arithmetic forward and so does the 2-step CFG loop (measured 6e-3 / 2e-2; distance to the reference's fp32 run unchanged)
  fused          in the DiT the RMSNorm + RoPE launch writes the e4m3 rows and scales itself; SVI_QK8_FUSED=0 (bf16 rows + quantiser launches): bit-identical
  default        the mode is off unless SVI_ATTN_QK8=1; with it off the attention's bits are those of the bf16 kernel
"""
import ctypes as C

import pytest
import torch

import synth
from gpu_util import dev, errs, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import svi_hip
    return svi_hip


class qk8:
    """with qk8(): the library's long-sequence attention runs in the fp8 QK^T mode"""
    def __init__(self, **more):
        self.sw = dict(SVI_ATTN_QK8=1, **more)

    def __enter__(self):
        from svi_hip import _lib as L
        for k, v in self.sw.items():
            L.set_switch(k, v)

    def __exit__(self, *a):
        from svi_hip import _lib as L
        for k in self.sw:
            L.set_switch(k, None)


def last_flagged():
    from svi_hip import _lib as L
    a, b = C.c_int32(), C.c_int32()
    L.check(L.lib().svi_attention_last_flagged(L.current_stream(), C.byref(a), C.byref(b)), "svi_attention_last_flagged")
    return a.value, b.value


def dequantised(x: torch.Tensor, heads: int) -> torch.Tensor:
    """[1, L, heads * 128] bf16 on the GPU -> the values the MX e4m3 operand holds (fp32, CPU)"""
    from oracle import mx8_oracle as mx
    xc = x[0].float().cpu()
    q, e = mx.mx8_quantize(xc)
    return mx.mx8_dequantize(q, e)


def sdpa64(q, k, v, heads, rows=None):
    L_, D = q.shape
    qh = q.double().view(L_, heads, 128).transpose(0, 1)
    if rows is not None:
        qh = qh[:, rows]
    kh = k.double().view(k.shape[0], heads, 128).transpose(0, 1)
    vh = v.double().view(v.shape[0], heads, 128).transpose(0, 1)
    p = torch.softmax(qh @ kh.transpose(1, 2) / 128 ** 0.5, dim=-1)
    return (p @ vh).transpose(0, 1).reshape(-1, D)


def operands(seed, Lq, Lk, heads, spread=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    D = heads * 128
    # channel-dependent magnitudes (what RMSNorm gains leave behind) so that the 32-channel blocks get different scales
    mag = torch.exp(torch.randn((D,), generator=g, device="cuda") * spread)
    q = (torch.randn((1, Lq, D), generator=g, device="cuda") * mag).to(torch.bfloat16)
    k = (torch.randn((1, Lk, D), generator=g, device="cuda") * mag.flip(0)).to(torch.bfloat16)
    v = torch.randn((1, Lk, D), generator=g, device="cuda").to(torch.bfloat16)
    return q, k, v


@pytest.mark.parametrize("Lq,Lk,heads", [(2304, 2304, 2), (4133, 4133, 3), (300, 2111, 1), (2048, 6000, 2)])
def test_kernel_equals_attention_over_the_dequantised_operands(hip, Lq, Lk, heads):
    q, k, v = operands(70 + heads, Lq, Lk, heads)
    base = hip.flash_attention(q, k, v, heads)
    with qk8():
        got = hip.flash_attention(q, k, v, heads)
        flagged, nwg = last_flagged()
        again = hip.flash_attention(q, k, v, heads)
    assert torch.equal(hip.flash_attention(q, k, v, heads), base)                   # the switch is off again: the bf16 kernel's bits
    assert nwg == ((Lq + 255) // 256) * heads, (flagged, nwg)      # (some of these operand sets are peaked enough to flag workgroups: the complete fp8 kernel recomputes them)
    assert torch.equal(got, again)
    qd, kd = dequantised(q, heads), dequantised(k, heads)
    want = sdpa64(qd, kd, v[0].float().cpu(), heads)
    exact = sdpa64(q[0].float().cpu(), k[0].float().cpu(), v[0].float().cpu(), heads)
    r, mx_, _ = errs(got[0], want)
    r_b = errs(base[0], exact)[0]
    dist = errs(got[0], base[0])[0]
    report("attn_qk8_kernel", Lq=Lq, Lk=Lk, heads=heads, flagged=flagged, workgroups=nwg, vs_fp64_over_dequantised=r, max_abs=mx_, bf16_kernel_vs_fp64=r_b, qk8_vs_bf16_kernel=dist,
           quantisation_alone=errs(want, exact)[0])
    assert torch.isfinite(got.float()).all() and r < 6e-3, (r, r_b)
    q_alone = errs(want, exact)[0]
    assert abs(dist - q_alone) < 0.1 * q_alone + 6e-3, (dist, q_alone)     # the whole distance from the bf16 kernel is what quantising Q and K costs on these operands


def test_second_pass_of_the_fp8_kernel(hip):
    """A late giant key every row projects on: every workgroup of the optimistic fp8 pass raises its flag and the complete fp8 kernel
    (tracked maximum, deferred rescale) recomputes it."""
    heads, L_ = 2, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn((1, L_, heads * 128), generator=g, device="cuda") + 2.0
    k = torch.randn((1, L_, heads * 128), generator=g, device="cuda")
    v = torch.randn((1, L_, heads * 128), generator=g, device="cuda")
    k[:, 3000] = 6.0                  # raw score 6 * 2 * 128 -> 196 log2 units: beyond the 160 the optimistic pass covers
    q, k, v = (a.to(torch.bfloat16).contiguous() for a in (q, k, v))
    with qk8():
        got = hip.flash_attention(q, k, v, heads)
        flagged, nwg = last_flagged()
    with qk8(SVI_FLASH_TWO_PASS=0):
        single = hip.flash_attention(q, k, v, heads)
    assert nwg == (L_ // 256) * heads and flagged == nwg, (flagged, nwg)
    want = sdpa64(dequantised(q, heads), dequantised(k, heads), v[0].float().cpu(), heads)
    r = errs(got[0], want)[0]
    report("attn_qk8_second_pass", flagged=flagged, workgroups=nwg, vs_fp64_over_dequantised=r, two_pass_vs_single_pass=errs(got, single)[0])
    assert torch.isfinite(got.float()).all() and r < 6e-3, r
    assert torch.equal(got, single)                       # every workgroup recomputed by the complete kernel: the single complete pass's bits


def test_key_axis_cut_into_pieces(hip):
    """SVI_FLASH_SPLIT=2: every work item cut into two pieces along the key axis (what a sequence-parallel rank's launch does to the items of its
    partly filled last round); the pieces' scale words and e4m3 rows start at the piece's first key."""
    heads, Lq, Lk = 2, 512, 8300
    q, k, v = operands(31, Lq, Lk, heads)
    with qk8():
        whole = hip.flash_attention(q, k, v, heads)
    with qk8(SVI_FLASH_SPLIT=2):
        got = hip.flash_attention(q, k, v, heads)
    want = sdpa64(dequantised(q, heads), dequantised(k, heads), v[0].float().cpu(), heads)
    r, rw = errs(got[0], want)[0], errs(got, whole)[0]
    report("attn_qk8_split", vs_fp64_over_dequantised=r, pieces_vs_whole=rw)
    assert torch.isfinite(got.float()).all() and r < 6e-3 and rw < 3e-3, (r, rw)


def test_c2_size_sampled_rows(hip):
    """The headline shape (L = 32760, 12 heads): 64 sampled query rows against fp64 over the dequantised operands."""
    L2, heads = 21 * 30 * 52, 12
    q, k, v = operands(9, L2, L2, heads, spread=0.5)
    with qk8():
        got = hip.flash_attention(q, k, v, heads)
    base = hip.flash_attention(q, k, v, heads)
    rows = torch.tensor(sorted(set([0, 1, 255, 256, 16383, L2 - 257, L2 - 1] + list(range(11, L2, L2 // 56)))))
    qd, kd = dequantised(q, heads), dequantised(k, heads)
    want = sdpa64(qd, kd, v[0].float().cpu(), heads, rows=rows)
    r = errs(got[0, rows.cuda()], want)[0]
    dist = errs(got, base)[0]
    report("attn_qk8_c2", rows=len(rows), vs_fp64_over_dequantised=r, qk8_vs_bf16_kernel=dist)
    assert torch.isfinite(got.float()).all() and r < 6e-3, (r, dist)


def test_forward_at_depth_stays_within_the_stated_distance(hip, golden):
    """dit_depth.npz (30-layer 1.3B, 7800 tokens, made by the reference): the forward with fp8 QK^T in every block's self-attention against the same
    forward in bf16 arithmetic, and both against the reference's fp32 run."""
    g = golden("dit_depth.npz")
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    f, h, w = synth.DEPTH_GRID
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **cfg).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(cfg), **cfg)
    del sd
    noise = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=1, device="cpu", dtype=torch.float32)
    pos = dev(torch.from_numpy(synth.text_context(seed + 1, 512, cfg["text_dim"], 64)))
    neg = dev(torch.from_numpy(synth.text_context(seed + 2, 512, cfg["text_dim"], 64)))
    sch = hip.FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(synth.DEPTH_STEPS, shift=5.0)
    base = m.forward(dev(noise), sch.timesteps[:1], pos)
    lat_base = hip.DenoiseLoop(m).sample(dev(noise), pos, neg, num_inference_steps=synth.DEPTH_STEPS, cfg_scale=5.0, sigma_shift=5.0)
    with qk8():
        got = m.forward(dev(noise), sch.timesteps[:1], pos)
        flagged, nwg = last_flagged()
        lat = hip.DenoiseLoop(m).sample(dev(noise), pos, neg, num_inference_steps=synth.DEPTH_STEPS, cfg_scale=5.0, sigma_shift=5.0)
    with qk8(SVI_QK8_FUSED=0):          # bf16 q | k written, two quantiser launches in front of every attention call: the same bits
        unfused = m.forward(dev(noise), sch.timesteps[:1], pos)
    assert torch.equal(got, unfused)
    r = errs(got, base)[0]
    rl = errs(lat, lat_base)[0]
    report("dit_depth_qk8", fwd_qk8_vs_bf16_arithmetic=r, fwd_qk8_vs_ref_fp32=errs(got[0], g["fwd_fp32"])[0], fwd_bf16_vs_ref_fp32=errs(base[0], g["fwd_fp32"])[0],
           loop_qk8_vs_bf16_arithmetic=rl, loop_qk8_vs_ref_fp32=errs(lat[0], g["lat_fp32"])[0], loop_bf16_vs_ref_fp32=errs(lat_base[0], g["lat_fp32"])[0],
           flagged=flagged, workgroups=nwg)
    assert torch.isfinite(got.float()).all() and flagged == 0
    assert r < 5e-2 and rl < 5e-2, (r, rl)


def test_a_kept_loop_recaptures_when_the_switch_moves(hip):
    """The mode changes the arithmetic, so a step graph recorded under the other setting must not be replayed: a DenoiseLoop kept across
    _lib.set_switch() calls re-captures (the switch-reload count is part of its key) and gives the eager loop's bits under each setting."""
    c = dict(dim=512, in_dim=16, ffn_dim=1024, out_dim=16, text_dim=64, freq_dim=256, patch_size=(1, 2, 2), num_layers=2, has_image_input=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(900, **c).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=4, **c)
    f, h, w = 4, 16, 32                                                   # 2048 tokens: the long-sequence kernel
    x = dev(synth.randn(901, 1, 16, f, 2 * h, 2 * w))
    cp, cn = dev(synth.text_context(902, 24, 64, 17)), dev(synth.text_context(903, 24, 64, 9))
    t = torch.tensor([712.5], device="cuda")
    loop, eager = hip.DenoiseLoop(m, graph=True), hip.DenoiseLoop(m, graph=False)

    def both():
        a, b = x.clone(), x.clone()
        for _ in range(2):
            loop.step(a, t, -0.03, cp, cn, 5.0)
            eager.step(b, t, -0.03, cp, cn, 5.0)
        return a, b
    try:
        a0, b0 = both()
        assert torch.equal(a0, b0)
        with qk8():
            a1, b1 = both()
        assert torch.equal(a1, b1) and not torch.equal(a1, a0)
        a2, b2 = both()
        assert torch.equal(a2, b2) and torch.equal(a2, a0)
        report("attn_qk8_loop_recapture", qk8_vs_bf16_two_steps=errs(a1, a0)[0])
    finally:
        loop.drop_graph()
