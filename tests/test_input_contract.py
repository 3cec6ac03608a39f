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


def test_teacache_with_the_cfg_pair_is_refused():
    """TeaCache + sequence parallelism is served (as the reference allows it); TeaCache on a CFG pair is not: DenoiseLoop.step must not
    take the TeaCache branch silently and run both full forwards on every rank."""
    import svi_hip
    m = _dit(synth.TINY_DIT)
    lat, ctx = torch.zeros(1, 16, 3, 8, 12), torch.zeros(1, 20, 64)
    tea = svi_hip.TeaCache(4, 0.1, "Wan2.1-T2V-1.3B")
    loop = svi_hip.DenoiseLoop(m, cfg_pair=object())
    with pytest.raises(NotImplementedError):
        loop.step(lat, torch.tensor([500.0]), -0.1, ctx, ctx, 5.0, tea_cache_posi=tea, tea_cache_nega=tea)
