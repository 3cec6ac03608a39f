"""-m gpu: sequence-parallel (Ulysses) forward, SURVEY §8e axis 3.  P shards run in one process on one GPU with the all-to-all
simulated by in-process gathers (svi_hip.sequence_parallel.forward_local): every kernel, row offset and head-group layout of the
multi-rank path is exercised; only the transport differs (tests/test_dist_gloo.py covers that over gloo).  Expected: the same
bits as the single-rank forward — each row / head sees the same operands in the same order."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import dev

pytestmark = pytest.mark.gpu

WIDE_T2V = dict(dim=512, in_dim=16, ffn_dim=1024, out_dim=16, text_dim=64, freq_dim=256, patch_size=(1, 2, 2), num_layers=2, has_image_input=False)
WIDE_I2V = dict(dim=512, in_dim=36, ffn_dim=768, out_dim=16, text_dim=64, freq_dim=256, patch_size=(1, 2, 2), num_layers=2, has_image_input=True)


def handles(hip, c, seed, n):
    sd = {k: torch.from_numpy(v).to("cuda", torch.bfloat16).contiguous() for k, v in synth.dit_state_dict(seed, **c).items()}
    out = []
    for _ in range(n):
        m = hip.WanDiT(eps=1e-6, num_heads=synth.num_heads_of(c), **c)
        m.bind(sd)                                       # all shards borrow the same weights
        out.append(m)
    return out


# 4 heads: ranks x head groups; grids: 48 tokens (ragged tiles), 2048 (the long-sequence attention kernel), 210 (odd shard lengths)
@pytest.mark.parametrize("P,G,grid", [(P, G, g) for g in [(2, 4, 6), (4, 16, 32), (3, 5, 14)]
                                      for P, G in [(1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (4, 1)] if (g[0] * g[1] * g[2]) % P == 0])
def test_sequence_parallel_is_bit_identical_t2v(P, G, grid):
    import svi_hip
    from svi_hip import sequence_parallel as sp
    f, h, w = grid
    ms = handles(svi_hip, WIDE_T2V, 900, P + 1)
    x = dev(synth.randn(901, 1, 16, f, 2 * h, 2 * w))
    ctx = dev(synth.text_context(902, 24, 64, 17))
    t = torch.tensor([712.5])
    want = ms[-1].forward(x, t, ctx)
    got = sp.forward_local(ms[:P], x, t, ctx, groups=G)
    assert got.shape == want.shape and torch.isfinite(got.float()).all()
    assert torch.equal(got, want)
    again = sp.forward_local(ms[:P], x, t, ctx, groups=G)            # second forward reuses the exchange buffers
    assert torch.equal(again, want) and len(ms[0]._sp_buffers) == 1


@pytest.mark.parametrize("P,G,grid", [(P, G, g) for g in [(2, 4, 6), (4, 16, 32), (3, 5, 14)]
                                      for P, G in [(1, 1), (2, 1), (2, 2), (4, 1)] if (g[0] * g[1] * g[2]) % P == 0])
def test_stacked_cfg_pair_on_sequence_shards_is_bit_identical(P, G, grid):
    """The CFG pair STACKED inside every sequence shard (svi_dit_sp_begin_pair / forward_local_pair: 2 L / P rows per rank, the conditional branch's on
    top; attention and the exchange blocks per branch; no CFG exchange between ranks) against the single-rank WanDiT.forward_cfg_pair and against two
    separate single-rank forwards: the same bits.  Shard lengths that are multiples of 8 (2048 / P), even (48 / P = 12 at P = 4) and odd (210 / 2 = 105:
    the unconditional branch's V^T columns then start at an odd column of the piece); RoPE positions of the lower half = the shard's own rows."""
    import svi_hip
    from svi_hip import sequence_parallel as sp
    f, h, w = grid
    ms = handles(svi_hip, WIDE_T2V, 900, P + 1)
    x = dev(synth.randn(901, 1, 16, f, 2 * h, 2 * w))
    ca, cb = dev(synth.text_context(902, 24, 64, 17)), dev(synth.text_context(903, 24, 64, 9))
    t = torch.tensor([712.5])
    want_a, want_b = ms[-1].forward(x, t, ca).clone(), ms[-1].forward(x, t, cb).clone()
    for m in ms:
        m.context_cache(True)
    try:
        one_a, one_b = ms[-1].forward_cfg_pair(x, t, ca, cb)
        assert torch.equal(one_a, want_a) and torch.equal(one_b, want_b)
        got_a, got_b = sp.forward_local_pair(ms[:P], x, t, ca, cb, groups=G)
        assert got_a.shape == want_a.shape and torch.isfinite(got_a.float()).all() and torch.isfinite(got_b.float()).all()
        assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)
        again = sp.forward_local_pair(ms[:P], x, t, ca, cb, groups=G)          # second step: cached prompts, reused buffers
        assert torch.equal(again[0], want_a) and torch.equal(again[1], want_b)
        assert torch.equal(sp.forward_local(ms[:P], x, t, ca, groups=G), want_a)      # the unstacked shard forward beside it (own buffers) is undisturbed
    finally:
        for m in ms:
            m.context_cache(False)
    with pytest.raises(RuntimeError):                     # without the context cache the stacked pair is refused, not silently run some other way
        sp.forward_local_pair(ms[:P], x, t, ca, cb, groups=G)


def test_stacked_cfg_pair_on_shards_i2v():
    import svi_hip
    from svi_hip import sequence_parallel as sp
    f, h, w = 3, 4, 4
    ms = handles(svi_hip, WIDE_I2V, 910, 3)
    x = dev(synth.randn(911, 1, 16, f, 2 * h, 2 * w))
    y = dev(synth.randn(912, 1, 20, f, 2 * h, 2 * w))
    clip = dev(synth.randn(913, 1, 257, 1280))
    ca, cb = dev(synth.text_context(915, 16, 64, 9)), dev(synth.text_context(916, 16, 64, 4))
    t = torch.tensor([92.5926])
    want = [ms[-1].forward(x, t, c, clip_feature=clip, y=y).clone() for c in (ca, cb)]
    for m in ms:
        m.context_cache(True)
    try:
        got = sp.forward_local_pair(ms[:2], x, t, ca, cb, clip_feature=clip, y=y)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    finally:
        for m in ms:
            m.context_cache(False)


@pytest.mark.parametrize("P,Dp,Ls", [(2, 128, 24), (4, 256, 105), (3, 128, 7), (2, 384, 8190)])
def test_unpack_kernels_follow_the_layout_algebra(P, Dp, Ls):
    """svi_sp_unpack_vt / svi_sp_unpack_out against their statement in tensor algebra (sequence_parallel.unpack_vt / unpack_out):
    shard lengths that are multiples of 8, even, and odd take the three copy widths."""
    from svi_hip import _lib as L
    from svi_hip import sequence_parallel as sp
    lds, L8 = (Ls + 7) // 8 * 8, (P * Ls + 7) // 8 * 8
    recv = dev(synth.randn(7, P, Dp, lds))
    out = torch.zeros((Dp, L8), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().svi_sp_unpack_vt(recv.data_ptr(), out.data_ptr(), P, Dp, Ls, lds, L8, 1, Dp, L.current_stream()))
    assert torch.equal(out, sp.unpack_vt(recv, Ls))
    for G in (1, 2):
        Dg = Dp // G
        r2 = dev(synth.randn(8, G, P, Ls * Dg))
        o2 = torch.empty((Ls, P * Dp), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().svi_sp_unpack_out(r2.data_ptr(), o2.data_ptr(), P, G, Ls, Dg, 1, L.current_stream()))
        assert torch.equal(o2, sp.unpack_out(r2, Ls))
        # the stacked CFG pair (nb = 2): a piece's columns hold branch b's tokens at [b Ls, (b + 1) Ls); out rows are [group][branch][Dg]; output pieces
        # [G][P][Ls][2][Dg] -> attn with the unconditional rows below the conditional ones
        lds2 = (2 * Ls + 7) // 8 * 8
        rv = dev(synth.randn(9, P, Dp, lds2))
        ov = torch.zeros((G * 2 * Dg, L8), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().svi_sp_unpack_vt(rv.data_ptr(), ov.data_ptr(), P, Dp, Ls, lds2, L8, 2, Dg, L.current_stream()))
        assert torch.equal(ov.view(G, 2, Dg, L8), sp.unpack_vt_pair(rv, Ls, G))
        r3 = dev(synth.randn(10, G, P, Ls * 2 * Dg))
        o3 = torch.empty((2 * Ls, P * Dp), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().svi_sp_unpack_out(r3.data_ptr(), o3.data_ptr(), P, G, Ls, Dg, 2, L.current_stream()))
        assert torch.equal(o3, sp.unpack_out_pair(r3, Ls))                                      # [b][row][src][g][c]


def test_sequence_parallel_i2v_and_add_condition():
    import svi_hip
    from svi_hip import sequence_parallel as sp
    f, h, w = 3, 4, 4
    ms = handles(svi_hip, WIDE_I2V, 910, 3)
    x = dev(synth.randn(911, 1, 16, f, 2 * h, 2 * w))
    y = dev(synth.randn(912, 1, 20, f, 2 * h, 2 * w))
    clip = dev(synth.randn(913, 1, 257, 1280))
    addc = dev(0.1 * synth.randn(914, 1, f * h * w, 512))
    ctx = dev(synth.text_context(915, 16, 64, 9))
    t = torch.tensor([92.5926])
    want = ms[-1].forward(x, t, ctx, clip_feature=clip, y=y, add_condition=addc)
    got = sp.forward_local(ms[:2], x, t, ctx, clip_feature=clip, y=y, add_condition=addc)
    assert torch.equal(got, want)


@pytest.mark.parametrize("P,grid,exact", [(3, (3, 4, 6), False), (3, (6, 16, 16), True), (2, (2, 4, 6), False), (8, (4, 16, 32), True)])
def test_gather_mode_serves_head_counts_that_do_not_divide(P, grid, exact):
    """4 heads over 3 or 8 ranks: the reference's USP falls back to ring attention there; here K / V^T are all-gathered and every rank
    attends with its own query rows of all heads.  Same kernels, rows regrouped into wavefronts from the shard's first row: within the
    attention tolerance (3e-3 on the forward), bit-identical when the shard length is a multiple of 256."""
    import svi_hip
    from svi_hip import sequence_parallel as sp
    f, h, w = grid
    ms = handles(svi_hip, WIDE_T2V, 900, P + 1)
    x = dev(synth.randn(901, 1, 16, f, 2 * h, 2 * w))
    ctx = dev(synth.text_context(902, 24, 64, 17))
    t = torch.tensor([712.5])
    want = ms[-1].forward(x, t, ctx)
    got = sp.forward_local(ms[:P], x, t, ctx, mode=None if 4 % P else "gather")
    assert got.shape == want.shape and torch.isfinite(got.float()).all()
    rel = float((got.float() - want.float()).norm() / want.float().norm())
    assert rel < 3e-3, rel
    if exact:
        assert (f * h * w // P) % 256 == 0 and torch.equal(got, want)


def test_shard_refuses_bad_divisions():
    import svi_hip
    from svi_hip import sequence_parallel as sp
    m = handles(svi_hip, WIDE_T2V, 900, 1)[0]
    with pytest.raises(ValueError):
        sp.SequenceShard(m, 0, 3, mode="ulysses")        # 4 heads over 3 ranks (the default would pick the gather mode)
    assert sp.SequenceShard(m, 0, 3).mode == "gather"
    sh = sp.SequenceShard(m, 0, 4)
    with pytest.raises(ValueError):                      # 2*3*5 = 30 tokens over 4 ranks
        sh.begin(dev(synth.randn(1, 1, 16, 2, 6, 10)), torch.tensor([500.0]), dev(synth.text_context(2, 8, 64, 5)))


def _dist_worker(rank, world, port, queue):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the test box; gloo moves
    try:                                                                # the exchanges through the host, the kernels run on the GPU
        import svi_hip
        torch.cuda.set_device(0)
        m = handles(svi_hip, WIDE_T2V, 900, 1)[0]
        x = dev(synth.randn(901, 1, 16, 2, 8, 12))
        ctx = dev(synth.text_context(902, 24, 64, 17))
        t = torch.tensor([712.5])
        want = m.forward(x, t, ctx)
        got = svi_hip.model_fn_wan_video(m, x, t, ctx, use_unified_sequence_parallel=True)     # -> forward_distributed
        loop = svi_hip.DenoiseLoop(m, sequence_parallel=True)
        lat = x.clone()
        loop.step(lat, t.cuda(), -0.05, ctx, dev(synth.text_context(903, 24, 64, 11)), 5.0)
        ref = x.clone()
        svi_hip.DenoiseLoop(m).step(ref, t.cuda(), -0.05, ctx, dev(synth.text_context(903, 24, 64, 11)), 5.0)
        # with the context cache on (as DenoiseLoop.sample leaves it) the sequence-parallel step STACKS the CFG pair on every rank
        # (forward_distributed_pair: no second forward, no CFG exchange): still the single-rank bits
        neg = dev(synth.text_context(903, 24, 64, 11))
        m.context_cache(True)
        try:
            lat2 = x.clone()
            loop.step(lat2, t.cuda(), -0.05, ctx, neg, 5.0)
            stacked_ok = bool(torch.equal(lat2, ref)) and loop.last_sp_form == "stacked pair"
        finally:
            m.context_cache(False)
        # TeaCache + sequence parallelism (allowed by the reference, svi_video.py:112-131): same skip pattern and the same bits as the
        # single-rank TeaCache loop; a huge threshold makes every middle step a skip, step 0 and the last step compute
        outs = {}
        for usp in (False, True):
            tc = svi_hip.TeaCache(5, 1e9, "Wan2.1-T2V-1.3B")
            xs, pattern = x.clone(), []
            for i in range(5):
                o = svi_hip.model_fn_wan_video(m, xs, torch.tensor([500.0 + 0.01 * i]).cuda(), ctx, tea_cache=tc, use_unified_sequence_parallel=usp)
                pattern.append(tc.previous_residual.shape[0])
                xs = (xs.float() + 0.05 * o.float()).to(torch.bfloat16)
            outs[usp] = (xs, tuple(pattern))
        ok_tea = bool(torch.equal(outs[False][0], outs[True][0])) and outs[True][1] == (x.shape[2] * (x.shape[3] // 2) * (x.shape[4] // 2) // world,) * 5
        queue.put((rank, bool(torch.equal(got, want)) and ok_tea, bool(torch.equal(lat, ref)) and stacked_ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_forward_distributed_across_processes(world):
    """model_fn_wan_video(use_unified_sequence_parallel=True) and a sequence-parallel DenoiseLoop step with a real process
    group (one process per rank, all-to-all / all-gather through torch.distributed): every rank gets the single-rank bits."""
    from spawn_util import run_ranks
    res = run_ranks(_dist_worker, world, timeout=300)
    assert sorted(r for r, _, _ in res) == list(range(world)) and all(a and b for _, a, b in res), res
