"""Host-side contract of the DiT entry points: what the reference rejects with a shape error must not reach the C side,
which walks raw pointers.  Handles are host objects until the first compute call, so this runs without a GPU."""
import pytest
import torch

import synth


def _dit(cfg):
    import svi_hip
    return svi_hip.WanDiT(eps=1e-6, num_heads=synth.num_heads_of(cfg), **cfg)


def test_t2v_input_contract():
    m = _dit(synth.TINY_DIT)
    x = torch.zeros(1, 16, 3, 8, 12)
    ctx = torch.zeros(1, 20, 64)
    m.check_inputs(x, (ctx,))
    with pytest.raises(ValueError, match="16"):
        m.check_inputs(torch.zeros(1, 15, 3, 8, 12), (ctx,))
    with pytest.raises(ValueError, match="patch"):
        m.check_inputs(torch.zeros(1, 16, 3, 7, 12), (ctx,))
    with pytest.raises(ValueError, match="text_dim"):
        m.check_inputs(x, (torch.zeros(1, 20, 65),))
    with pytest.raises(ValueError, match="context"):
        m.check_inputs(x, (torch.zeros(2, 20, 64),))
    with pytest.raises(ValueError, match="in_dim == 16"):
        m.check_inputs(x, (ctx,), y=torch.zeros(1, 20, 3, 8, 12))
    with pytest.raises(ValueError, match="add_condition"):
        m.check_inputs(x, (ctx,), add_condition=torch.zeros(1, 71, 128))
    m.check_inputs(x, (ctx,), add_condition=torch.zeros(1, 72, 128))


def test_i2v_input_contract():
    m = _dit(synth.TINY_DIT_I2V)
    x, ctx = torch.zeros(2, 16, 2, 8, 8), torch.zeros(2, 16, 64)
    y, clip = torch.zeros(2, 20, 2, 8, 8), torch.zeros(2, 257, 1280)
    m.check_inputs(x, (ctx, ctx), clip, y)
    with pytest.raises(ValueError, match="takes y"):
        m.check_inputs(x, (ctx,), clip, None)
    with pytest.raises(ValueError, match="takes y"):
        m.check_inputs(x, (ctx,), clip, torch.zeros(2, 19, 2, 8, 8))
    with pytest.raises(ValueError, match="clip_feature"):
        m.check_inputs(x, (ctx,), None, y)
    with pytest.raises(ValueError, match="clip_feature"):
        m.check_inputs(x, (ctx,), torch.zeros(2, 256, 1280), y)


def test_forward_checks_before_touching_the_device():
    m = _dit(synth.TINY_DIT)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.forward(torch.zeros(1, 16, 3, 8, 12), torch.tensor([1.0]), torch.zeros(1, 20, 64))


def test_teacache_on_a_cfg_pair_runs_this_ranks_branch_with_its_own_cache():
    """Round 6 (was a refusal): on a CFG pair DenoiseLoop.step hands the pair ONE forward — this rank's branch through model_fn_wan_video with that branch's
    TeaCache — instead of running both branches on every rank; without a cache for its branch it refuses with a message."""
    import svi_hip
    m = _dit(synth.TINY_DIT)
    lat, ctx = torch.zeros(1, 16, 3, 8, 12), torch.zeros(1, 20, 64)
    tea = svi_hip.TeaCache(4, 0.1, "Wan2.1-T2V-1.3B")
    seen = {}

    class Pair:
        role = 1

        def step(self, forward, cfg_step, latents, timestep, dsigma, ctx_pos, ctx_neg, cfg_scale, uncond_overrides=None, **cond):
            seen.update(forward=forward, cfg_step=cfg_step, scale=cfg_scale)
            return latents
    loop = svi_hip.DenoiseLoop(m, cfg_pair=Pair())
    assert loop.step(lat, torch.tensor([500.0]), -0.1, ctx, ctx, 5.0, tea_cache_posi=tea, tea_cache_nega=tea) is lat
    assert callable(seen["forward"]) and seen["scale"] == 5.0
    with pytest.raises(ValueError, match="one TeaCache per branch"):
        loop.step(lat, torch.tensor([500.0]), -0.1, ctx, ctx, 5.0, tea_cache_posi=tea, tea_cache_nega=None)
