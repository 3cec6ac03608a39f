"""Multi-process (gloo, world_size 2, CPU) tests of the N>1 path: clip sharding, the end-of-round all-gather and the
stitching rule.  The property under test: the gathered window is identical for any number of ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
from svi_hip import parallel  # noqa: E402


from spawn_util import run_ranks as _run_ranks  # noqa: E402


def _clip(k):
    """Stand-in for a denoised clip that depends only on k — like seed = k*42 does (test_svi.py:425)."""
    g = torch.Generator("cpu").manual_seed(parallel.clip_seed(k))
    return torch.randn((16, 3, 4, 4), generator=g).to(torch.bfloat16)


_clip.proto = torch.zeros((16, 3, 4, 4), dtype=torch.bfloat16)


def _worker(rank, world, port, num_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = parallel.ClipParallel()
        mine = par.my_clips(num_clips)
        window = parallel.run_window(_clip, num_clips, par)
        tails = par.all_gather_motion_tails({k: _clip(k) for k in mine}, num_clips, 1) if mine else None
        q.put((rank, mine, [w.float() for w in window], None if tails is None else [t.float() for t in tails]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_clips", [2, 5])
def test_two_ranks_reproduce_single_rank_window(num_clips):
    res = _run_ranks(_worker, 2, num_clips)
    serial = [_clip(k).float() for k in range(num_clips)]
    seen = []
    for rank, mine, window, tails in res:
        assert mine == [k for k in range(num_clips) if k % 2 == rank]
        seen += mine
        assert len(window) == num_clips
        for k in range(num_clips):
            assert torch.equal(window[k], serial[k])          # bit-identical to the 1-rank order
        if tails is not None:
            for k in range(num_clips):
                assert torch.equal(tails[k], serial[k][:, -1:])
    assert sorted(seen) == list(range(num_clips))


def _toy_forward(x, t, ctx, **kw):
    """Stand-in for the DiT forward: any deterministic function of (latents, timestep, context)."""
    return (torch.tanh(x.float() * 0.7 + ctx.float().mean() + t.float() * 1e-3) * 0.9).to(torch.bfloat16)


def _ref_cfg_step(lat, c, u, s, dsigma):
    """svi_video.py:410 + flow_match.py:53-64 with the reference's bf16 rounding points."""
    v = (u + s * (c - u))
    lat.copy_(lat + v * dsigma)


def _serial_cfg_clip(seed, steps=4):
    g = torch.Generator("cpu").manual_seed(seed)
    lat = torch.randn((1, 16, 3, 4, 4), generator=g).to(torch.bfloat16)
    cp, cn = torch.randn((1, 8, 16), generator=g).to(torch.bfloat16), torch.randn((1, 8, 16), generator=g).to(torch.bfloat16)
    return lat, cp, cn


def _cfg_worker(rank, world, port, num_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pair, pidx, npairs = parallel.CfgPair.split_world()
        out = {}
        for k in parallel.shard_units(num_clips, pidx, npairs):
            lat, cp, cn = _serial_cfg_clip(parallel.clip_seed(k))
            for i in range(4):
                pair.step(_toy_forward, _ref_cfg_step, lat, torch.tensor([900.0 - 100 * i]), -0.1, cp, cn, 5.0)
            out[k] = lat.float()
        q.put((rank, pair.role, pidx, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_clips", [(2, 1), (4, 3)])
def test_cfg_pair_matches_serial_order(world, num_clips):
    """cond on one rank, uncond on the other, one all-gather per step: both ranks of a pair end with the latents the
    serial loop produces, bit for bit; with 4 ranks the pairs additionally shard the clips."""
    res = _run_ranks(_cfg_worker, world, num_clips)
    serial = {}
    for k in range(num_clips):
        lat, cp, cn = _serial_cfg_clip(parallel.clip_seed(k))
        for i in range(4):
            t = torch.tensor([900.0 - 100 * i])
            _ref_cfg_step(lat, _toy_forward(lat, t, cp), _toy_forward(lat, t, cn), 5.0, -0.1)
        serial[k] = lat.float()
    covered = set()
    for rank, role, pidx, out in res:
        assert role == rank % 2 and pidx == rank // 2
        assert sorted(out) == [k for k in range(num_clips) if k % (world // 2) == pidx]
        for k, v in out.items():
            assert torch.equal(v, serial[k])
            covered.add(k)
    assert covered == set(range(num_clips))


def _sp_worker(rank, world, port, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svi_hip import sequence_parallel as sp
        Lfull, D, G = 12 * world, 256 * world, 2
        Ls, Dp = Lfull // world, D // world
        Dg = Dp // G
        g = torch.Generator("cpu").manual_seed(5)
        Q, K, V = [torch.randn((Lfull, D), generator=g).to(torch.bfloat16) for _ in range(3)]     # the global tensors, known to all
        rows = slice(rank * Ls, (rank + 1) * Ls)
        ldvt = (Ls + 7) // 8 * 8
        vt = torch.zeros((D, ldvt), dtype=torch.bfloat16)               # what svi_dit_sp_block_qkv leaves on this rank: V^T ...
        vt[:, :Ls] = V[rows].t()
        qs, ks = sp.send_layout_qk(Q[rows], world, G), sp.send_layout_qk(K[rows], world, G)   # ... and q | k in send order
        a2a = lambda t: sp.all_to_all(t.contiguous())                    # noqa: E731
        vt_full = sp.unpack_vt(a2a(vt.reshape(world, Dp * ldvt)).reshape(world, Dp, ldvt), Ls)
        cols = slice(rank * Dp, (rank + 1) * Dp)
        ok_qkv = torch.equal(vt_full[:, :Lfull], V[:, cols].t()) and bool((vt_full[:, Lfull:] == 0).all())
        O = (Q.float() * 0.5 + K.float()).to(torch.bfloat16)             # any [L, D] result of "attention", column block = head group
        o_send = []
        for gi in range(G):
            gc = slice(rank * Dp + gi * Dg, rank * Dp + (gi + 1) * Dg)
            q, k = a2a(qs[gi]).reshape(Lfull, Dg), a2a(ks[gi]).reshape(Lfull, Dg)    # what arrives is the token-major operand itself
            ok_qkv = ok_qkv and torch.equal(q, Q[:, gc]) and torch.equal(k, K[:, gc])
            o_send.append(O[:, gc].reshape(world, Ls * Dg))              # the attention output [L, Dg]: contiguous per destination
        attn = sp.unpack_out(torch.stack([a2a(o) for o in o_send]), Ls)
        queue.put((rank, ok_qkv, torch.equal(attn, O[rows])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sequence_parallel_exchange_layout(world):
    """tokens -> heads and back over a real all-to-all (gloo): after the first exchange a rank holds ALL tokens of ITS head
    group (q, k row-major per head group without any unpacking, V transposed, zero padded), after the second its OWN rows of all heads."""
    res = _run_ranks(_sp_worker, world)
    assert sorted(r for r, _, _ in res) == list(range(world))
    assert all(a and b for _, a, b in res), res


def _sp_pair_worker(rank, world, port, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svi_hip import sequence_parallel as sp
        Lfull, D, G = 12 * world, 256 * world, 2
        Ls, Dp = Lfull // world, D // world
        Dg = Dp // G
        g = torch.Generator("cpu").manual_seed(6)
        Q, K, V = [[torch.randn((Lfull, D), generator=g).to(torch.bfloat16) for _ in range(2)] for _ in range(3)]     # [branch] global tensors, known to all
        rows = slice(rank * Ls, (rank + 1) * Ls)
        stack = lambda t: torch.cat([t[0][rows], t[1][rows]])                # noqa: E731  the rank's rows of the pair: conditional branch on top
        lds2 = (2 * Ls + 7) // 8 * 8
        vt = torch.zeros((D, lds2), dtype=torch.bfloat16)                     # what the rank's V projection leaves: V^T of its 2 Ls rows
        vt[:, :2 * Ls] = stack(V).t()
        qs, ks = sp.send_layout_qk_pair(stack(Q), world, G), sp.send_layout_qk_pair(stack(K), world, G)
        a2a = lambda t: sp.all_to_all(t.contiguous())                        # noqa: E731
        vt_full = sp.unpack_vt_pair(a2a(vt.reshape(world, Dp * lds2)).reshape(world, Dp, lds2), Ls, G)      # [G, 2, Dg, L8]
        ok = bool((vt_full[..., Lfull:] == 0).all())
        O = [(Q[b].float() * 0.5 + K[b].float()).to(torch.bfloat16) for b in range(2)]      # any [L, D] "attention" result per branch
        o_send = []
        for gi in range(G):
            gc = slice(rank * Dp + gi * Dg, rank * Dp + (gi + 1) * Dg)
            q, k = a2a(qs[gi]).reshape(Lfull, 2, Dg), a2a(ks[gi]).reshape(Lfull, 2, Dg)   # token-major [L, 2 Dg]: the branches are twice as many heads
            for b in range(2):
                ok = ok and torch.equal(q[:, b], Q[b][:, gc]) and torch.equal(k[:, b], K[b][:, gc]) and torch.equal(vt_full[gi, b, :, :Lfull], V[b][:, gc].t())
            o_send.append(torch.stack([O[0][:, gc], O[1][:, gc]], dim=1).reshape(world, Ls * 2 * Dg))      # the launch's output [L, 2, Dg]: contiguous per destination
        attn = sp.unpack_out_pair(torch.stack([a2a(o) for o in o_send]), Ls)
        queue.put((rank, ok, torch.equal(attn, stack(O))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_stacked_cfg_pair_exchange_layout(world):
    """The stacked CFG pair on sequence shards (forward_distributed_pair) over a real all-to-all (gloo): a rank sends its 2 Ls rows with the two branches side
    by side per token, so after the first exchange it holds, per head group, ALL tokens as a token-major [L, 2 Dg] operand (the unconditional branch = more
    heads of the same attention launch) and V^T as [group][branch][Dg] rows; after the second exchange its own rows of both branches, conditional on top.
    No exchange BETWEEN the branches anywhere: the CFG combination needs none."""
    res = _run_ranks(_sp_pair_worker, world)
    assert sorted(r for r, _, _ in res) == list(range(world))
    assert all(a and b for _, a, b in res), res


def _groups_worker(rank, world, port, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pair, sp_group, S = parallel.split_cfg_sequence()
        a = torch.tensor([float(2 ** rank)])
        dist.all_reduce(a, group=sp_group)                   # sum of 2^r over the members identifies the group
        b = torch.tensor([float(2 ** rank)])
        dist.all_reduce(b, group=pair.group)
        queue.put((rank, pair.role, S, int(a.item()), int(b.item())))
    finally:
        dist.destroy_process_group()


def test_cfg_x_sequence_group_layout():
    """world 4 = 2 CFG branches x 2 sequence shards: ranks {0,1} / {2,3} are the sequence groups, {0,2} / {1,3} the CFG pairs."""
    res = {r[0]: r[1:] for r in _run_ranks(_groups_worker, 4)}
    assert res == {0: (0, 2, 3, 5), 1: (0, 2, 3, 10), 2: (1, 2, 12, 5), 3: (1, 2, 12, 10)}


def test_single_process_paths():
    par = parallel.ClipParallel()
    assert (par.rank, par.world) == (0, 1) and par.my_clips(3) == [0, 1, 2]
    w = parallel.run_window(_clip, 3, par)
    assert all(torch.equal(w[k], _clip(k)) for k in range(3))
    assert parallel.shard_units(8, 3, 8) == [3] and parallel.shard_units(3, 5, 8) == []
    assert sorted(sum((parallel.shard_units(11, r, 4) for r in range(4)), [])) == list(range(11))


def test_seed_and_prompt_schedule_match_reference_loop():
    assert [parallel.clip_seed(k) for k in range(3)] == [0, 42, 84]          # test_svi.py:425, seed_times=42
    assert parallel.clip_seed(7, -1) is None
    assert [parallel.clip_prompt_index(k, 3, 2) for k in range(8)] == [0, 0, 1, 1, 2, 2, 0, 0]
    assert parallel.clip_prompt_index(5, 3, 1, use_first_prompt_only=True) == 0


def test_stitching_rule():
    clips = [list(range(10 * i, 10 * i + 5)) for i in range(3)]
    assert parallel.stitch_window(clips, 1) == [0, 1, 2, 3, 10, 11, 12, 13, 20, 21, 22, 23, 24]   # test_svi.py:472-476
    assert parallel.stitch_window(clips[:1], 1) == clips[0]
    assert len(parallel.stitch_window(clips, 0)) == 15
