"""TeaCache (pipelines/svi_video.py:23-72, 114-130): host decision logic on CPU, the skip/update/store data path on the GPU.

Golden: tests/golden/teacache_tiny.npz — the reference's own TeaCache class and model_fn_wan_video (compiled out of
pipelines/svi_video.py by tests/gen_golden.py) driving the reference WanModel for 8 steps: per-step outputs, which steps
skipped the blocks, and the t_mod tensors the decisions were taken on."""
import numpy as np
import pytest
import torch

import synth


def test_host_logic_reproduces_reference_decisions(golden):
    """svi_hip.TeaCache.check on the reference's own t_mod sequence takes the reference's decisions (accumulate, threshold,
    reset, first/last step always computed, wrap-around of the step counter)."""
    from svi_hip.teacache import TeaCache
    g = golden("teacache_tiny.npz")
    tc = TeaCache(synth.TEA_STEPS, synth.TEA_THRESH, synth.TEA_MODEL)
    tm = torch.from_numpy(g["t_mod"]).to(torch.bfloat16)
    for rounds in range(2):                                     # second round: the counter wrapped, same decisions again
        got = [tc.check(None, None, tm[i]) for i in range(synth.TEA_STEPS)]
        assert got == [bool(v) for v in g["skipped"]]
    assert any(g["skipped"][1:-1]) and not all(g["skipped"][1:-1])   # the fixture exercises both branches
    with pytest.raises(ValueError):
        TeaCache(8, 0.1, "not-a-model")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["svi_hip", "duck"])
def test_teacache_loop_matches_reference(golden, which):
    """8-step loop through model_fn_wan_video(..., tea_cache=...): same skip pattern as the reference, outputs within the
    whole-forward tolerance on computed steps and on skipped steps (x + cached residual -> head)."""
    import svi_hip
    from gpu_util import dev, errs, report
    g = golden("teacache_tiny.npz")
    c, grid, nt, nv, seed = synth.TINY_DIT, (3, 4, 6), 20, 13, 100
    f, h, w = grid
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    m = svi_hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    x = dev(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    ctx = dev(synth.text_context(seed + 2, nt, c["text_dim"], nv))
    if which == "svi_hip":
        tc = svi_hip.TeaCache(synth.TEA_STEPS, synth.TEA_THRESH, synth.TEA_MODEL)
    else:
        # an object with the REFERENCE class's interface and bookkeeping (check / previous_residual / previous_hidden_states),
        # as the unchanged pipeline would hand over: model_fn drives it through check() only
        class RefLike(svi_hip.TeaCache):
            def check(self, dit, x, t_mod):
                skip = super().check(dit, x, t_mod)
                if not skip:
                    self.previous_hidden_states = x.clone()
                return skip
        tc = RefLike(synth.TEA_STEPS, synth.TEA_THRESH, synth.TEA_MODEL)
    skipped, worst = [], 0.0
    for i in range(synth.TEA_STEPS):
        ts = torch.tensor([synth.TEA_TIMESTEPS[i]], dtype=torch.float32, device="cuda")
        before = tc.previous_residual
        calls_before = None if before is None else before.clone()
        o = svi_hip.model_fn_wan_video(m, x, ts, ctx, tea_cache=tc)
        was_skip = calls_before is not None and torch.equal(calls_before, tc.previous_residual) and i not in (0, synth.TEA_STEPS - 1)
        skipped.append(bool(was_skip))
        r, mx, _ = errs(o, g["outs"][i])
        worst = max(worst, r)
        assert o.shape == g["outs"][i].shape and r < 2e-2, (i, r)
        x = (x.float() + 0.05 * o.float()).to(torch.bfloat16)      # the generator's latent update
    report("teacache_loop", which=which, worst_rel_l2=worst, skipped=str(skipped))
    assert skipped == [bool(v) for v in g["skipped"]]
    # t_mod seam used for the decisions: close to the reference's bf16 t_mod
    tm = m.time_mod(torch.tensor([synth.TEA_TIMESTEPS[3]]))
    r, _, _ = errs(tm, g["t_mod"][3])
    assert r < 1e-2, r


@pytest.mark.gpu
def test_denoise_loop_with_teacache_runs_fewer_block_stacks():
    import svi_hip
    from gpu_util import dev
    c, grid, nt, nv, seed = synth.TINY_DIT, (3, 4, 6), 20, 13, 100
    f, h, w = grid
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    m = svi_hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    lat = dev(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    cp, cn = dev(synth.text_context(seed + 2, nt, c["text_dim"], nv)), dev(synth.text_context(seed + 3, nt, c["text_dim"], nv))
    loop = svi_hip.DenoiseLoop(m)
    plain = loop.sample(lat, cp, cn, num_inference_steps=6)
    tea0 = loop.sample(lat, cp, cn, num_inference_steps=6, tea_cache_l1_thresh=-1e30, tea_cache_model_id=synth.TEA_MODEL)
    assert torch.equal(plain, tea0)                              # a threshold nothing can stay below: every step computes -> same bits
    tea = loop.sample(lat, cp, cn, num_inference_steps=6, tea_cache_l1_thresh=1e30, tea_cache_model_id=synth.TEA_MODEL)
    assert torch.isfinite(tea.float()).all() and not torch.equal(tea, plain)    # everything between first and last step skipped
    with pytest.raises(ValueError):
        loop.sample(lat, cp, cn, num_inference_steps=6, tea_cache_l1_thresh=0.1, tea_cache_model_id="")


@pytest.mark.gpu
def test_teacache_on_a_cfg_pair_gives_the_serial_loops_bits():
    """TeaCache on a CFG pair (round 6): each rank runs its branch through model_fn_wan_video with that branch's cache, the pair exchanges noise_pred, both
    apply CFG + Euler.  Emulated in one process — two DenoiseLoops over one WanDiT, a pair object per role whose `step` keeps its branch's prediction until the
    partner's arrives (what the all-gather does) — against the serial TeaCache loop: the two caches take the same decisions step for step (they are functions
    of the time modulation), so the latents are the serial loop's, bit for bit, and skipped steps really skip on both 'ranks'."""
    import svi_hip
    from svi_hip import ops
    from gpu_util import dev
    c, grid, nt, nv, seed = synth.TINY_DIT, (3, 4, 6), 20, 13, 100
    f, h, w = grid
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    m = svi_hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    lat0 = dev(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    cp, cn = dev(synth.text_context(seed + 2, nt, c["text_dim"], nv)), dev(synth.text_context(seed + 3, nt, c["text_dim"], nv))
    steps, thr = 8, 0.15
    serial = svi_hip.DenoiseLoop(m, graph=False)
    serial.scheduler.set_timesteps(steps, shift=5.0)
    tp, tn = svi_hip.TeaCache(steps, thr, synth.TEA_MODEL), svi_hip.TeaCache(steps, thr, synth.TEA_MODEL)
    want = lat0.clone()
    ts = serial.scheduler.timesteps.to("cuda", torch.float32)
    for i, t in enumerate(serial.scheduler.timesteps):
        serial.step(want, ts[i:i + 1], serial.scheduler.step_delta(t), cp, cn, 5.0, tea_cache_posi=tp, tea_cache_nega=tn)

    mailbox = {}

    class Pair:
        def __init__(self, role):
            self.role = role

        def step(self, forward, cfg_step, latents, timestep, dsigma, ctx_pos, ctx_neg, cfg_scale, uncond_overrides=None, **cond):
            mailbox[self.role] = (forward(latents, timestep, ctx_pos if self.role == 0 else ctx_neg, **cond).clone(), latents, cfg_scale, dsigma)
            if len(mailbox) == 2:                    # both branches are in: every rank applies the same update to its own copy of the latents
                for role in (0, 1):
                    cfg_step(mailbox[role][1], mailbox[0][0], mailbox[1][0], cfg_scale, dsigma)
                mailbox.clear()
            return latents
    loops = [svi_hip.DenoiseLoop(m, cfg_pair=Pair(r)) for r in (0, 1)]
    caches = [(svi_hip.TeaCache(steps, thr, synth.TEA_MODEL), svi_hip.TeaCache(steps, thr, synth.TEA_MODEL)) for _ in (0, 1)]
    lats = [lat0.clone(), lat0.clone()]
    for i, t in enumerate(serial.scheduler.timesteps):
        for r in (0, 1):
            loops[r].step(lats[r], ts[i:i + 1], serial.scheduler.step_delta(t), cp, cn, 5.0, tea_cache_posi=caches[r][0], tea_cache_nega=caches[r][1])
    assert torch.equal(lats[0], want) and torch.equal(lats[1], want)
    # rank 0 only ever touched its conditional cache, rank 1 its unconditional one, and both walked the serial caches' steps
    assert caches[0][0].step == tp.step and caches[1][1].step == tn.step and caches[0][1].step == 0 and caches[1][0].step == 0
    assert not torch.equal(want, svi_hip.DenoiseLoop(m).sample(lat0, cp, cn, num_inference_steps=steps))       # the threshold does skip steps here
