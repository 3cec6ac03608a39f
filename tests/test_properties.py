"""Size-independent properties of the host logic around the hot path (hypothesis; CPU only, no compute calls).

  * the two sequence-parallel exchanges (svi_hip/sequence_parallel.py; the reference's USP chunk / all-to-all / all-gather,
    pipelines/svi_video.py:119-135, distributed/xdit_context_parallel.py) are exact re-layouts for ANY shard count, ragged sizes
    included: tokens -> heads gives every rank all tokens of its head group, heads -> tokens is its inverse;
  * clip sharding is a partition, stitching follows test_svi.py:472-476 for any clip lengths;
  * FlowMatchScheduler (schedulers/flow_match.py:3-63) ladders are monotone and the 50 Euler deltas of a clip telescope to the
    sigma the clip started from.
"""
import math

import numpy as np
import torch

import synth
from hypothesis import given, settings
from hypothesis import strategies as st

from svi_hip import parallel
from svi_hip import sequence_parallel as sp
from svi_hip.scheduler import FlowMatchScheduler

FAST = settings(max_examples=60, deadline=None)


def _ids(L, D):
    """[L, D] tensor whose entry (l, c) encodes its own coordinates (exact in bf16 for the sizes used: values < 256)."""
    return (torch.arange(L)[:, None] % 16 * 16 + torch.arange(D)[None, :] % 16).to(torch.bfloat16)


@FAST
@given(P=st.sampled_from([1, 2, 3, 4, 6, 8]), G=st.sampled_from([1, 2, 3]), heads_per=st.integers(1, 2), Ls=st.integers(1, 21), seed=st.integers(0, 2 ** 16))
def test_tokens_to_heads_and_back(P, G, heads_per, Ls, seed):
    Dg = 8 * heads_per                       # stands in for heads_per x 128 columns; the layout code only sees column blocks
    Dp = G * Dg
    D, L = P * Dp, P * Ls
    g = torch.Generator().manual_seed(seed)
    Q, K, V = (torch.randn((L, D), generator=g).to(torch.bfloat16) for _ in range(3))
    ldvt = (Ls + 7) // 8 * 8
    qs, ks, vs = [], [], []
    for r in range(P):
        rows = slice(r * Ls, (r + 1) * Ls)
        qs.append(sp.send_layout_qk(Q[rows], P, G))          # [G, P, Ls*Dg]
        ks.append(sp.send_layout_qk(K[rows], P, G))
        vt = torch.zeros((D, ldvt), dtype=torch.bfloat16)
        vt[:, :Ls] = V[rows].t()
        vt[:, Ls:] = 7.0                     # whatever sits in the pad columns of a shard must not reach the attention operand
        vs.append(vt.reshape(P, Dp, ldvt))   # V^T as the projection writes it: row block j = rank j's piece
    qr, kr, vr = sp.all_to_all_local(qs), sp.all_to_all_local(ks), sp.all_to_all_local([v.reshape(P, Dp * ldvt) for v in vs])
    outs = []
    for j in range(P):
        cols = slice(j * Dp, (j + 1) * Dp)
        vt = sp.unpack_vt(vr[j].reshape(P, Dp, ldvt), Ls)
        assert vt.shape == (Dp, (L + 7) // 8 * 8)
        assert torch.equal(vt[:, :L], V[:, cols].t()) and not vt[:, L:].any()      # zero pad: the kernel reads those columns
        o = []
        for gi in range(G):
            gc = slice(j * Dp + gi * Dg, j * Dp + (gi + 1) * Dg)
            q, k = qr[j][gi].reshape(L, Dg), kr[j][gi].reshape(L, Dg)              # what arrived IS the token-major operand
            assert torch.equal(q, Q[:, gc]) and torch.equal(k, K[:, gc])
            o.append((q.float() + 2 * k.float()).to(torch.bfloat16).reshape(P, Ls * Dg))   # "attention" of the group, [L, Dg]
        outs.append(torch.stack(o))          # [G, P, Ls*Dg]: contiguous per destination
    back = sp.all_to_all_local(outs)
    want = (Q.float() + 2 * K.float()).to(torch.bfloat16)
    for r in range(P):
        attn = sp.unpack_out(back[r], Ls)
        assert attn.shape == (Ls, D) and torch.equal(attn, want[r * Ls:(r + 1) * Ls])


@FAST
@given(P=st.integers(1, 8), G=st.integers(1, 3), Ls=st.integers(1, 9), Dg=st.sampled_from([8, 16, 24]))
def test_exchange_moves_every_element_exactly_once(P, G, Ls, Dg):
    """Coordinates survive the round trip: nothing is duplicated or dropped by send layout / exchange / unpack."""
    Dp = G * Dg
    D, L = P * Dp, P * Ls
    X = _ids(L, D)
    back = sp.all_to_all_local([sp.send_layout_qk(X[r * Ls:(r + 1) * Ls], P, G) for r in range(P)])
    for j in range(P):
        for gi in range(G):
            assert torch.equal(back[j][gi].reshape(L, Dg), X[:, j * Dp + gi * Dg:j * Dp + (gi + 1) * Dg])
    packs = [torch.stack([X[:, j * Dp + gi * Dg:j * Dp + (gi + 1) * Dg].reshape(P, Ls * Dg) for gi in range(G)]) for j in range(P)]
    got = torch.cat([sp.unpack_out(b, Ls) for b in sp.all_to_all_local(packs)], dim=0)
    assert torch.equal(got, X)


@FAST
@given(heads=st.integers(1, 40), tokens=st.integers(1, 80000))
def test_head_group_choice(heads, tokens):
    g = sp.head_groups(heads, tokens)
    assert 1 <= g <= heads and heads % g == 0
    if g > 1:
        assert (heads // g) * ((tokens + 255) // 256) >= 256          # a group still fills the chip


@FAST
@given(n=st.integers(0, 200), world=st.integers(1, 16))
def test_clip_sharding_is_a_balanced_partition(n, world):
    shards = [parallel.shard_units(n, r, world) for r in range(world)]
    flat = sorted(k for s in shards for k in s)
    assert flat == list(range(n))
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) <= 1
    assert all(s == sorted(s) for s in shards)                      # a rank meets its clips in window order
    assert all(k % world == r for r, s in enumerate(shards) for k in s)


@FAST
@given(lengths=st.lists(st.integers(1, 30), min_size=1, max_size=8), m=st.integers(0, 5))
def test_stitching_rule(lengths, m):
    """test_svi.py:472-476: every clip but the last loses its final `num_motion_frames` frames."""
    clips = [[(i, f) for f in range(n)] for i, n in enumerate(lengths)]
    out = parallel.stitch_window(clips, m)
    want = []
    for i, c in enumerate(clips):
        want += c if (i == len(clips) - 1 or m == 0) else c[:-m] if m < len(c) else []
    assert out == want
    assert out[-lengths[-1]:] == clips[-1]                           # the last clip is whole
    assert len(out) == sum(max(n - m, 0) for n in lengths[:-1]) + lengths[-1]


@FAST
@given(k=st.integers(0, 500), n=st.integers(1, 12), rep=st.integers(1, 7), first=st.booleans(), seed_times=st.sampled_from([-1, 0, 1, 42, 1000]))
def test_prompt_and_seed_schedule(k, n, rep, first, seed_times):
    """test_svi.py:425-438: seed = chunk_idx * seed_times (-1 = unseeded); prompt index cycles every `rep` clips."""
    want = 0 if first else (k // rep) % n
    assert parallel.clip_prompt_index(k, n, rep, first) == want
    s = parallel.clip_seed(k, seed_times)
    assert s is None if seed_times == -1 else s == k * seed_times


@FAST
@given(steps=st.integers(1, 120), shift=st.floats(1.0, 12.0), strength=st.floats(0.05, 1.0))
def test_flow_match_ladder(steps, shift, strength):
    sch = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)     # as SVIVideoPipeline builds it (svi_video.py:148)
    sch.set_timesteps(steps, denoising_strength=strength, shift=shift)
    sig = sch.sigmas.double()
    assert len(sig) == steps and torch.equal(sch.timesteps, sch.sigmas * 1000)
    assert bool((sig[1:] < sig[:-1]).all()) and sig[-1] > 0 and sig[0] <= 1.0 + 1e-6
    assert math.isclose(float(sig[0]), shift * strength / (1 + (shift - 1) * strength), rel_tol=1e-5)
    deltas = [sch.step_delta(t) for t in sch.timesteps]
    assert all(d < 0 for d in deltas)
    assert math.isclose(sum(deltas), -float(sig[0]), rel_tol=0, abs_tol=1e-5)   # the clip ends at sigma = 0
    assert math.isclose(deltas[-1], -float(sig[-1]), abs_tol=1e-7)


@FAST
@given(steps=st.integers(2, 60), i=st.integers(0, 59))
def test_nearest_timestep_lookup(steps, i):
    """step() finds its position by nearest timestep (flow_match.py:54-55): a timestep that went through bf16 or a device round
    trip still lands on its own step."""
    sch = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(steps)
    i = i % steps
    t = sch.timesteps[i]
    for noisy in (t, t.to(torch.bfloat16).float() if steps <= 30 else t, t + 1e-3):
        want = float((sch.sigmas[i + 1] if i + 1 < steps else 0.0) - sch.sigmas[i])
        assert math.isclose(sch.step_delta(noisy), want, abs_tol=1e-7)


# ---------------------------------------------------------------------------------------------------- prompt-side encoders, checkpoints
@settings(max_examples=60, deadline=None)
@given(st.sampled_from([8, 16, 32, 64]), st.sampled_from([32, 64, 128, 256]), st.integers(1, 300))
def test_relative_bucket_table_of_the_library_equals_the_oracle_for_any_configuration(num_buckets, max_dist, length):
    """svi_t5_relative_buckets (host code of libsvi_hip) against the restatement of T5RelativeEmbedding._relative_position_bucket that is
    pinned to the reference's own table at (32, 128, 512): same buckets for every offset, monotone in |offset|, mirrored with the sign bit."""
    from oracle import encoders_oracle as eo
    from svi_hip import encoders
    if max_dist <= num_buckets // 4:
        return
    got = np.array(encoders.relative_position_buckets(num_buckets, max_dist, length), np.int32)
    want = eo.relative_position_buckets(num_buckets, max_dist, length)
    assert np.array_equal(got, want)
    zero = length - 1
    assert got[zero] == 0 and got.max() <= num_buckets - 1
    neg, pos = got[:zero][::-1], got[zero + 1:]
    assert np.array_equal(pos, neg + num_buckets // 2) and np.all(np.diff(pos) >= 0)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.sampled_from([16, 36]), st.integers(1, 5), st.integers(1, 4), st.booleans(), st.booleans())
def test_dit_configuration_is_recovered_from_shapes(heads, in_dim, layers, ffn_mult, image, talk):
    """svi_hip.checkpoint.infer_dit_config on the key -> shape table of any WanModel configuration (the table is the reference's own:
    tests/test_reference_keys.py checks synth.dit_param_shapes against WanModel.state_dict())."""
    from svi_hip.checkpoint import infer_dit_config
    cfg = dict(dim=128 * heads, in_dim=in_dim, ffn_dim=256 * ffn_mult, out_dim=16, text_dim=64, freq_dim=256, patch_size=(1, 2, 2), num_layers=layers,
               has_image_input=image)
    if talk:
        cfg["enable_multitalk"] = True
    got = infer_dit_config(synth.dit_param_shapes(**cfg))
    want = dict(cfg, num_heads=heads, eps=1e-6)
    assert got == want
