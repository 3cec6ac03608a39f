"""-m gpu: DiT block / whole forward / denoise loop on the HIP path against the golden vectors of the
reference and against the oracle.

Stated tolerances (rel-L2 over the whole output tensor):
  block   vs oracle(bf16 rounding) <= 6e-3 ;  vs reference fp32 golden <= 1.5e-2
  forward vs reference bf16 golden  <= 2e-2  (and not worse than 2x the reference's own bf16-vs-fp32 gap)
  4-step CFG denoise loop vs reference fp32 golden <= 5e-2   (SURVEY §8c suggested bounds)
"""
import numpy as np
import pytest
import torch

import synth
from gpu_util import bf16r, dev, errs, host, report
from oracle import flow_match_oracle as fmo
from oracle import wan_dit_oracle as wdo
from test_oracle_dit import CASES, inputs, make_cfg

pytestmark = pytest.mark.gpu


def build(hip, c, seed):
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    return m, {k: bf16r(v) for k, v in sd.items()}


@pytest.fixture(scope="module")
def hip():
    import svi_hip
    return svi_hip


@pytest.mark.parametrize("name", list(CASES))
def test_block_forward(hip, golden, name):
    c, grid, nt, nv, ts, seed = CASES[name]
    f, h, w = grid
    g = golden(f"dit_{name}.npz")
    m, sdb = build(hip, c, seed)
    L = f * h * w
    bx = torch.from_numpy(synth.randn(seed + 5, 1, L, c["dim"]))
    bctx = torch.from_numpy(synth.randn(seed + 6, 1, nt + (257 if c["has_image_input"] else 0), c["dim"]))
    btm = torch.from_numpy(0.5 * synth.randn(seed + 7, 1, 6, c["dim"]))
    got = m.block_forward(0, dev(bx), dev(bctx), dev(btm), grid)
    cfg = make_cfg(c)
    want_b = wdo.dit_block(sdb, "blocks.0.", bf16r(bx), bf16r(bctx), bf16r(btm), wdo.rope_table_3d(128, grid), cfg, "bf16")
    r_or, mx_or, _ = errs(got, want_b)
    r_ref32, _, _ = errs(got, g["block0_fp32"])
    r_ref16, _, _ = errs(got, g["block0_bf16"])
    report("dit_block", case=name, vs_oracle_bf16=r_or, vs_ref_fp32=r_ref32, vs_ref_bf16=r_ref16, max_abs=mx_or)
    assert r_or < 6e-3 and r_ref32 < 1.5e-2, (r_or, r_ref32, r_ref16)


@pytest.mark.parametrize("name", list(CASES))
def test_forward(hip, golden, name):
    c, grid, nt, nv, ts, seed = CASES[name]
    g = golden(f"dit_{name}.npz")
    m, sdb = build(hip, c, seed)
    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    got = m.forward(dev(x), torch.tensor([ts]), dev(ctx), **{k: dev(v) for k, v in kw.items()})
    want_b = wdo.dit_forward(sdb, make_cfg(c), x, torch.tensor([ts]), ctx, rounding="bf16", **kw)
    r_or, mx, _ = errs(got, want_b)
    r_ref16, _, _ = errs(got, g["out_bf16"])
    r_ref32, _, _ = errs(got, g["out_fp32"])
    from conftest import rel_l2
    noise = rel_l2(g["out_bf16"], g["out_fp32"])
    report("dit_forward", case=name, vs_oracle_bf16=r_or, vs_ref_bf16=r_ref16, vs_ref_fp32=r_ref32, ref_bf16_vs_fp32=noise)
    assert got.shape == g["out_fp32"].shape
    assert r_ref16 < 2e-2 and r_ref32 < max(2e-2, 2 * noise), (r_or, r_ref16, r_ref32, noise)


def test_model_fn_signature_and_refusals(hip):
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    m, _ = build(hip, c, seed)
    x, ctx, _ = inputs(c, grid, nt, nv, seed)
    a = hip.model_fn_wan_video(m, dev(x), torch.tensor([ts]).cuda(), dev(ctx))
    b = m.forward(dev(x), torch.tensor([ts]), dev(ctx))
    assert torch.equal(a, b)                       # deterministic: same kernels, same order
    # USP without a process group is the plain forward, as in the reference (svi_video.py:119-121); with one: tests/test_gpu_sp.py
    c2 = hip.model_fn_wan_video(m, dev(x), torch.tensor([ts]).cuda(), dev(ctx), use_unified_sequence_parallel=True)
    assert torch.equal(c2, b)


def test_add_condition_and_batch(hip):
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    f, h, w = grid
    m, sdb = build(hip, c, seed)
    x, ctx, _ = inputs(c, grid, nt, nv, seed)
    add = torch.from_numpy(0.1 * synth.randn(9, 1, f * h * w, c["dim"]))
    got = m.forward(dev(x), torch.tensor([ts]), dev(ctx), add_condition=dev(add))
    want = wdo.dit_forward(sdb, make_cfg(c), x, torch.tensor([ts]), ctx, add_condition=bf16r(add), rounding="bf16")
    assert errs(got, want)[0] < 1e-2
    xb = torch.cat([x, x.flip(-1)]); cb = torch.cat([ctx, ctx])
    gb = m.forward(dev(xb), torch.tensor([ts, ts]), dev(cb))
    g0 = m.forward(dev(x), torch.tensor([ts]), dev(ctx))
    assert torch.equal(gb[0:1], g0)


def test_denoise_loop_matches_reference(hip, golden):
    g = golden("denoise_tiny.npz")
    c, seed, grid = synth.TINY_DIT, 300, (2, 4, 4)
    f, h, w = grid
    m, sdb = build(hip, c, seed)
    lat = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=11, device="cpu", dtype=torch.float32)
    pos = torch.from_numpy(synth.text_context(seed + 2, 16, c["text_dim"], 9))
    neg = torch.from_numpy(synth.text_context(seed + 3, 16, c["text_dim"], 4))
    loop = hip.DenoiseLoop(m)
    out = loop.sample(dev(lat), dev(pos), dev(neg), num_inference_steps=4, cfg_scale=5.0, sigma_shift=5.0)
    cfg = make_cfg(c)
    want_b = fmo.denoise_loop(lambda x, t, cx: wdo.dit_forward(sdb, cfg, x, t, cx, rounding="bf16"), lat, pos, neg, 4, 5.0, 5.0, "bf16")
    r_or = errs(out, want_b)[0]
    r_ref = errs(out, g["latents"])[0]
    report("denoise_loop", vs_oracle_bf16=r_or, vs_ref_fp32=r_ref)
    assert r_ref < 5e-2 and r_or < 3e-2, (r_or, r_ref)


def test_c1_scale_forward_vs_oracle(hip):
    """Config C1 geometry (17f 256x256 -> L=1280) at a 2-layer, 2-head width: multi-tile GEMMs and 20 key tiles."""
    c = dict(synth.SMALL_DIT)
    m, sdb = build(hip, c, 700)
    x = torch.from_numpy(synth.randn(701, 1, 16, 5, 32, 32))
    ctx = torch.from_numpy(synth.text_context(702, 512, c["text_dim"], 60))
    ts = torch.tensor([833.3333])
    got = m.forward(dev(x), ts, dev(ctx))
    want = wdo.dit_forward(sdb, make_cfg(c), x, ts, ctx, rounding="bf16")
    want32 = wdo.dit_forward({k: torch.from_numpy(v) for k, v in synth.dit_state_dict(700, **c).items()}, make_cfg(c), x, ts, ctx)
    r, mx, _ = errs(got, want)
    r32 = errs(got, want32)[0]
    report("dit_forward_c1grid", vs_oracle_bf16=r, vs_oracle_fp32=r32, max_abs=mx)
    assert r < 1e-2 and r32 < 2e-2, (r, r32)


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
def test_context_cache_is_bit_identical(hip, name):
    """svi_dit_context_cache: projected context + per-block cross K/V computed once per context pointer, same bits."""
    c, grid, nt, nv, ts, seed = CASES[name]
    m, _ = build(hip, c, seed)
    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    xd, cd = dev(x), dev(ctx)
    kwd = {k: dev(v) for k, v in kw.items()}
    ref = m.forward(xd, torch.tensor([ts]), cd, **kwd)
    ctx2 = dev(torch.from_numpy(synth.randn(seed + 77, *ctx.shape)))
    ref2 = m.forward(xd, torch.tensor([ts]), ctx2, **kwd)
    m.context_cache(True)
    a1 = m.forward(xd, torch.tensor([ts]), cd, **kwd)          # fills entry 1
    b1 = m.forward(xd, torch.tensor([ts]), ctx2, **kwd)        # fills entry 2
    a2 = m.forward(xd, torch.tensor([ts * 0.5]), cd, **kwd)    # reuses entry 1 at another timestep
    a3 = m.forward(xd, torch.tensor([ts]), cd, **kwd)          # reuses entry 1
    m.context_cache(False)
    want_half = m.forward(xd, torch.tensor([ts * 0.5]), cd, **kwd)
    assert torch.equal(a1, ref) and torch.equal(a3, ref) and torch.equal(b1, ref2) and torch.equal(a2, want_half)


def test_context_cache_survives_prompt_turnover(hip):
    """A rolling window hands the DiT a new prompt embedding per clip: the old tensor dies, the allocator may return its address
    for the next one, or the pipeline may overwrite it in place.  The cache is keyed by pointer on the C side, so WanDiT pins what
    the cache has seen and drops it on an in-place write: results must always be those of the uncached forward."""
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    m, _ = build(hip, c, seed)
    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    xd, t = dev(x), torch.tensor([ts])
    others = [dev(torch.from_numpy(synth.randn(seed + 200 + i, *ctx.shape))) for i in range(12)]
    want0 = m.forward(xd, t, dev(ctx)).clone()
    wants = [m.forward(xd, t, o).clone() for o in others]
    m.context_cache(True)
    try:
        cur = dev(ctx)
        assert torch.equal(m.forward(xd, t, cur), want0)
        del cur                                            # clip over: the embedding is dropped by its owner ...
        for o, w in zip(others, wants):                    # ... and fresh ones of the same shape arrive (more than the pin capacity)
            fresh = o.clone()
            assert torch.equal(m.forward(xd, t, fresh), w)
            assert torch.equal(m.forward(xd, t, fresh), w)         # served from the cache
            del fresh
        keep = dev(ctx)
        assert torch.equal(m.forward(xd, t, keep), want0)
        keep.copy_(others[3])                              # in-place overwrite of a cached embedding
        assert torch.equal(m.forward(xd, t, keep), wants[3])
        with pytest.raises(ValueError):                    # a conversion would hand the cache a temporary
            m.forward(xd, t, keep.float())
    finally:
        m.context_cache(False)
    assert torch.equal(m.forward(xd, t, keep.float()), wants[3])


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
@pytest.mark.parametrize("cache", [False, True])
def test_cfg_pair_is_bit_identical_to_two_forwards(hip, name, cache):
    """svi_dit_forward_cfg_pair shares the prompt-independent head of the forward (time embedding, patchify, block 0's
    self-attention) between the cond and uncond forwards of a step: same bits as two svi_dit_forward calls."""
    c, grid, nt, nv, ts, seed = CASES[name]
    m, _ = build(hip, c, seed)
    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    kw = {k: dev(v) for k, v in kw.items()}
    cp, cn = dev(ctx), dev(-np.asarray(ctx))
    t = torch.tensor([ts])
    a = m.forward(dev(x), t, cp, **kw).clone()
    b = m.forward(dev(x), t, cn, **kw).clone()
    m.context_cache(cache)
    try:
        for _ in range(2):                         # second round is served from the context cache when it is on
            pa, pb = m.forward_cfg_pair(dev(x), t, cp, cn, **kw)
            assert torch.equal(pa, a) and torch.equal(pb, b)
    finally:
        m.context_cache(False)
    assert not torch.equal(a, b)


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
def test_graph_replay_is_bit_identical(hip, name):
    """DenoiseLoop(graph=True): the two forwards of a step captured once into a hipGraph and replayed with a refreshed device
    timestep — the same kernels on the same operands, so the same bits as the eager loop, over a whole (short) clip and a second clip
    (new latents / prompts: re-capture)."""
    c, grid, nt, nv, ts, seed = CASES[name]
    f, h, w = grid
    m, _ = build(hip, c, seed)
    _, ctx, kw = inputs(c, grid, nt, nv, seed)
    kw = {k: dev(v) for k, v in kw.items()}
    for clip in range(2):
        lat = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=5 + clip, device="cpu", dtype=torch.float32)
        cp, cn = dev(np.asarray(ctx) * (1 + clip)), dev(-np.asarray(ctx))
        want = hip.DenoiseLoop(m, graph=False).sample(dev(lat), cp, cn, num_inference_steps=5, cfg_scale=5.0, **kw)
        loop = hip.DenoiseLoop(m, graph=True)
        got = loop.sample(dev(lat), cp, cn, num_inference_steps=5, cfg_scale=5.0, **kw)
        assert torch.equal(got, want)
        assert loop._graph is None                  # the graph reads the clip's tensors and cache entries: it dies with the clip


def test_graph_never_replays_against_stale_state(hip):
    """What a captured step has baked in (addresses of latents / prompts / outputs, a context-cache hit, workspace and cache-entry
    pointers) must still hold at every replay.  One DenoiseLoop(graph=True) object is kept across clips whose tensors are freed and
    re-created (the caching allocator hands the new prompt the old prompt's address), across a context-cache reset, an interleaved
    eager forward of another size on the same handle (workspace re-layout) and an in-place prompt edit: every result equals the
    eager loop's bits."""
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    f, h, w = grid
    m, _ = build(hip, c, seed)
    _, ctx, _ = inputs(c, grid, nt, nv, seed)
    loop = hip.DenoiseLoop(m, graph=True)
    eager = hip.DenoiseLoop(m, graph=False)
    seen = set()
    for clip in range(3):
        lat = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=20 + clip, device="cpu", dtype=torch.float32)
        cp, cn = dev(np.asarray(ctx) * (1.0 + 0.5 * clip)), dev(-np.asarray(ctx))
        seen.add(cp.data_ptr())
        got = loop.sample(dev(lat), cp, cn, num_inference_steps=3, cfg_scale=5.0)
        want = eager.sample(dev(lat), cp, cn, num_inference_steps=3, cfg_scale=5.0)
        assert torch.equal(got, want), clip
        del cp, cn, got, want
    # step-level: the caller keeps the loop and its tensors, other things move underneath
    m.context_cache(True)
    try:
        lat0 = dev(hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=31, device="cpu", dtype=torch.float32))
        cp, cn = dev(np.asarray(ctx)), dev(-np.asarray(ctx))
        t = torch.tensor([ts], device="cuda")

        def both(tag):
            a, b = lat0.clone(), lat0.clone()
            for _ in range(2):
                loop.step(a, t, -0.03, cp, cn, 5.0)
                eager.step(b, t, -0.03, cp, cn, 5.0)
            assert torch.equal(a, b), tag
        both("first capture")
        gen0 = m.generation()
        big = dev(synth.randn(77, 1, 16, 3 * f, 2 * h, 2 * w))     # 3 x the tokens (the stacked CFG pair already lays out 2 x)
        m.forward(big, t, cp)                                     # another problem size on the same handle: the workspace is laid out anew
        assert m.generation() != gen0
        both("after a workspace re-layout")
        m.context_cache(False)
        m.context_cache(True)                                     # every cached projection is gone: a replay would read dead entries
        both("after a context-cache reset")
        cp.mul_(0.5)                                              # in-place prompt edit: same address, new contents
        both("after an in-place prompt edit")
    finally:
        loop.drop_graph()
        m.context_cache(False)


def test_first_graphed_step_and_stream_buffer_release(hip):
    """Round 4: the hipGraph is the single-rank default and the clip's FIRST step is the eager run on the loop's capture stream itself (its result is kept, not a
    discarded warm-up).  (a) Throw-away loops used for one step each, with other work enqueued right behind them, give the eager step's bits every time (a
    timestep copy that raced the capture stream made ~40 % of such first steps wrong before the fix); (b) DenoiseLoop.release() / svi_stream_buffers_release
    free the library's per-stream buffers, move svi_dit_generation, and the loop captures again and still matches."""
    from svi_hip import _lib as L
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    f, h, w = grid
    m, _ = build(hip, c, seed)
    _, ctx, _ = inputs(c, grid, nt, nv, seed)
    cp, cn = dev(np.asarray(ctx)), dev(-np.asarray(ctx))
    x = dev(hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=41, device="cpu", dtype=torch.float32))
    t = torch.tensor([ts], device="cuda")
    want = x.clone()
    hip.DenoiseLoop(m, graph=False).step(want, t, -0.04, cp, cn, 5.0)
    for i in range(12):
        got = x.clone()
        hip.DenoiseLoop(m).step(got, t, -0.04, cp, cn, 5.0)          # temporary loop: destroyed while its CFG / Euler kernel may still be queued
        m.forward(x, torch.tensor([500.0 + i]), cp)                    # more work on the same handle right behind
        assert torch.equal(got, want), i
    loop = hip.DenoiseLoop(m)
    a = x.clone()
    loop.step(a, t, -0.04, cp, cn, 5.0)
    loop.step(a, t, -0.04, cp, cn, 5.0)                                # a replay
    gen0 = m.generation()
    assert loop._graph is not None and loop._capture_stream is not None
    # long-sequence attention on the capture stream creates per-stream flag words there; a short model has none: create one explicitly
    with torch.cuda.stream(loop._capture_stream):
        q = dev(synth.randn(3, 1, 2304, 128))
        hip.flash_attention(q, q, q, 1)
    torch.cuda.synchronize()
    loop.release()
    assert loop._graph is None and loop._capture_stream is None and m.generation() != gen0
    b = x.clone()
    loop.step(b, t, -0.04, cp, cn, 5.0)
    loop.step(b, t, -0.04, cp, cn, 5.0)
    assert torch.equal(a, b)
    L.release_stream_buffers(None)                                     # every stream of the device: legal at any quiet point


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
def test_context_refill_recomputes_an_entry_in_place(hip, name):
    """svi_dit_context_refill (the rolling window's clip boundary): the next prompt (and CLIP feature) is written INTO the tensors a cache entry is
    keyed by; the entry is recomputed where it stands — same bits as an uncached forward on the new values, and neither epoch() nor generation()
    moves (a captured step graph stays valid)."""
    c, grid, nt, nv, ts, seed = CASES[name]
    m, _ = build(hip, c, seed)
    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    xd, t = dev(x), torch.tensor([ts])
    kwd = {k: dev(v) for k, v in kw.items()}
    ctx_a, ctx_b = dev(ctx), dev(torch.from_numpy(synth.randn(seed + 301, *ctx.shape)))
    ctx_b[:, nv - 2:] = 0                                  # another identical-suffix length: the tail summary must be recomputed too
    kwb = dict(kwd)
    if "clip_feature" in kwd:
        kwb["clip_feature"] = dev(torch.from_numpy(synth.randn(seed + 302, *kw["clip_feature"].shape)))
    want_a = m.forward(xd, t, ctx_a, **kwd).clone()
    want_b = m.forward(xd, t, ctx_b, **kwb).clone()
    assert not torch.equal(want_a, want_b)
    m.context_cache(True)
    try:
        slot, slot_kw = ctx_a.clone(), {k: v.clone() for k, v in kwd.items()}
        assert torch.equal(m.forward(xd, t, slot, **slot_kw), want_a)
        assert torch.equal(m.forward(xd, t, slot, **slot_kw), want_a)          # a hit
        e0, g0 = m.epoch(), m.generation()
        slot.copy_(ctx_b)
        if "clip_feature" in slot_kw:
            slot_kw["clip_feature"].copy_(kwb["clip_feature"])
        m.refill_context(slot, slot_kw.get("clip_feature"))
        assert (m.epoch(), m.generation()) == (e0, g0)
        assert torch.equal(m.forward(xd, t, slot, **slot_kw), want_b)          # served from the refilled entry
        assert (m.epoch(), m.generation()) == (e0, g0)
        slot.copy_(ctx_a)                                                      # an in-place write WITHOUT a refill is still caught (pins): never stale
        if "clip_feature" in slot_kw:
            slot_kw["clip_feature"].copy_(kwd["clip_feature"])
        assert torch.equal(m.forward(xd, t, slot, **slot_kw), want_a)
    finally:
        m.context_cache(False)


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
def test_resident_loop_replays_one_graph_across_clips(hip, name):
    """DenoiseLoop(resident=True): the loop owns the clip's device tensors, every clip's inputs are copied in, prompt entries are refilled in place, and
    the step graph captured by the first clip is what every later clip of the same shapes replays — no eager step, no re-capture.  Same bits as the
    eager loop for every clip; a clip of another shape captures anew; results do not alias the loop's tensors."""
    c, grid, nt, nv, ts, seed = CASES[name]
    f, h, w = grid
    m, _ = build(hip, c, seed)
    _, ctx, kw = inputs(c, grid, nt, nv, seed)
    loop = hip.DenoiseLoop(m, resident=True)
    eager = hip.DenoiseLoop(m, graph=False)
    assert loop.resident and loop.graph
    neg = dev(-np.asarray(ctx))

    def clip_inputs(clip):
        lat = dev(hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=60 + clip, device="cpu", dtype=torch.float32))
        cp = dev(np.asarray(ctx) * (1.0 + 0.25 * (clip % 3)))                   # a new prompt tensor per clip; the negative prompt is one tensor throughout
        return lat, cp, {k: dev(np.asarray(v) * (1.0 + 0.1 * clip)) for k, v in kw.items()}
    wants = []
    for clip in range(4):
        lat, cp, kwd = clip_inputs(clip)
        wants.append(eager.sample(lat, cp, neg, num_inference_steps=3, cfg_scale=5.0, **kwd))
    lat2 = dev(hip.generate_noise((1, 16, f, 4 * h, 2 * w), seed=70, device="cpu", dtype=torch.float32))
    kw2 = {k: (dev(synth.randn(seed + 9, 1, v.shape[1], f, 4 * h, 2 * w)) if k == "y" else dev(v)) for k, v in kw.items()}
    want2 = eager.sample(lat2, dev(ctx), neg, num_inference_steps=2, cfg_scale=5.0, **kw2)
    gots = []
    for clip in range(4):
        lat, cp, kwd = clip_inputs(clip)
        gots.append(loop.sample(lat, cp, neg, num_inference_steps=3, cfg_scale=5.0, **kwd))
        assert m._ctx_cache_on and torch.equal(gots[-1], wants[clip]), clip
        del lat, cp, kwd
    assert loop.captures == 1
    assert all(torch.equal(g, w) for g, w in zip(gots, wants))  # earlier results were not overwritten by later clips
    assert torch.equal(loop.sample(lat2, dev(ctx), neg, num_inference_steps=2, cfg_scale=5.0, **kw2), want2) and loop.captures == 2
    loop.close()
    assert not m._ctx_cache_on


@pytest.mark.parametrize("name", ["tiny_t2v", "tiny_i2v"])
def test_identical_trailing_context_rows_count_as_one_key(hip, name):
    """The prompter zero-fills a prompt embedding past the prompt's tokens and cross-attention attends to all rows without a mask:
    identical input rows -> identical K / V rows -> one key counted m times (csrc/svi_dit.hip ctx_tail_*).  (i) with a zero-padded
    context the forward with the shortcut equals the forward that walks every row (SVI_CROSS_DEDUP=0) to bf16 noise — and both meet
    the reference's golden output, which attends to all rows; (ii) a context without identical trailing rows gives the same BITS either
    way; (iii) rows identical to the last one in the MIDDLE of the context are not merged (only a suffix is)."""
    from svi_hip import _lib as L
    c, grid, nt, nv, ts, seed = CASES[name]
    g = np.load(f"{__import__('conftest').GOLDEN}/dit_{name}.npz")
    m, _ = build(hip, c, seed)
    x, ctx, kw = inputs(c, grid, nt, nv, seed)
    kw = {k: dev(v) for k, v in kw.items()}
    assert nv < nt and not np.asarray(ctx)[0, nv:].any()
    t = torch.tensor([ts])

    def fwd(context, dedup):
        L.set_switch("SVI_CROSS_DEDUP", 1 if dedup else 0)
        try:
            return m.forward(dev(x), t, dev(context), **kw).clone()
        finally:
            L.set_switch("SVI_CROSS_DEDUP", None)
    a, b = fwd(ctx, True), fwd(ctx, False)
    r = errs(a, b)[0]
    ra, rb = errs(a, g["out_bf16"])[0], errs(b, g["out_bf16"])[0]
    report("cross_dedup", case=name, dedup_vs_all_rows=r, dedup_vs_ref_bf16=ra, all_rows_vs_ref_bf16=rb, padded_rows=nt - nv)
    # the two differ by the bf16 rounding of P: bf16(m p) once against m times bf16(p) — the attention tolerance; each is as close to
    # the reference as the other (measured: 3.1e-3 apart, 3.7e-3 / 3.7e-3 from the reference's bf16 output)
    assert r < 6e-3 and ra < 2e-2 and rb < 2e-2 and abs(ra - rb) < 2e-3, (r, ra, rb)
    full = np.asarray(ctx).copy()
    full[0, nv:] = synth.randn(seed + 40, nt - nv, full.shape[-1])             # no two rows alike
    assert torch.equal(fwd(full, True), fwd(full, False))
    mid = full.copy()
    mid[0, 2] = mid[0, -1]                                                     # a twin of the last row far from the end
    assert torch.equal(fwd(mid, True), fwd(mid, False))


def test_stacked_pair_falls_back_once_when_its_workspace_does_not_fit(hip):
    """the stacked CFG pair needs a workspace of 2 L rows; when that allocation fails, forward_pair runs the unstacked form (same bits)
    — and REMEMBERS it: the next step neither frees and re-allocates (the handle's generation stands still) nor, on the capture pass that follows the eager
    step of a graphed loop, asks a capturing stream to grow the workspace (which was refused as INVALID, so the 14B-width case this targets could not run
    the graphed loop at all).  SVI_WS_LIMIT_MB turns a budget into the same SVI_ERR_OOM a failed hipMalloc gives."""
    from svi_hip import _lib as L
    c, seed = synth.SMALL_DIT, 150
    f, h, w = 4, 24, 32                                                # 3072 tokens: the workspace is MiBs, so a MiB budget separates L rows from 2 L
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    ctx = synth.text_context(seed + 2, 24, c["text_dim"], 24)
    cp, cn = dev(ctx), dev(-np.asarray(ctx))
    lat = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=9, device="cpu", dtype=torch.float32)

    def model():
        m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
        m.context_cache(True)
        return m

    want = hip.DenoiseLoop(model(), graph=False).sample(dev(lat), cp, cn, num_inference_steps=4, cfg_scale=5.0)     # no budget: the stacked pair
    try:
        # the smallest whole-MiB budget under which ONE forward of L rows still runs
        fit = None
        for mb in range(1, 200):
            L.set_switch("SVI_WS_LIMIT_MB", mb)
            try:
                model().forward(dev(lat), torch.tensor([500.0]), cp)
                fit = mb
                break
            except RuntimeError as e:
                assert "SVI_WS_LIMIT_MB" in str(e)
        assert fit is not None and fit >= 4, fit                        # (2 L rows then need nearly twice that: beyond the budget)
        m = model()
        with pytest.raises(RuntimeError):                              # the budget really refuses the doubled workspace
            m2 = model()
            m2.forward(torch.cat([dev(lat), dev(lat)], dim=3), torch.tensor([500.0]), cp)
        x = dev(lat).to(torch.bfloat16)
        t = torch.tensor([637.5], device="cuda")
        eager = hip.DenoiseLoop(m, graph=False)
        eager.step(x, t, -0.04, cp, cn, 5.0)                           # stacked refused -> unstacked, remembered
        g1 = m.generation()
        eager.step(x, t, -0.04, cp, cn, 5.0)
        assert m.generation() == g1                                    # no free + re-allocation per step any more
        for graph in (False, True):                                    # graph=True: the capture pass behind the eager first step succeeds
            got = hip.DenoiseLoop(m, graph=graph).sample(dev(lat), cp, cn, num_inference_steps=4, cfg_scale=5.0)
            assert torch.equal(got, want), graph
    finally:
        L.set_switch("SVI_WS_LIMIT_MB", None)
