"""I2V conditioning assembly (pipelines/svi_video.py:313-350): host logic on CPU, HIP VAE path on GPU."""
import numpy as np
import pytest
import torch

import synth
from oracle import wan_vae_oracle as wvo


def test_mask_layout_matches_reference_lines():
    from svi_hip.conditioning import condition_mask
    m = condition_mask(9, 16, 24, num_condition_frames=1, device="cpu")
    assert m.shape == (4, 3, 2, 3)
    # first latent frame = the first video frame repeated 4x -> all four channels one; everything else zero
    assert torch.all(m[:, 0] == 1) and torch.all(m[:, 1:] == 0)
    m = condition_mask(9, 16, 24, num_condition_frames=3, ref_pad_cfg=True, device="cpu")
    # video frames 0,1,2 conditioned: latent frame 0 (4 copies of frame 0) all ones; latent frame 1 = video frames 1..4 -> channels 0,1
    assert torch.all(m[:, 0] == 1)
    assert torch.all(m[0, 1] == 1) and torch.all(m[1, 1] == 1) and torch.all(m[2:, 1] == 0) and torch.all(m[:, 2] == 0)


@pytest.mark.parametrize("ref_pad_num", [0, 2, -1])
def test_condition_video_padding_rules(ref_pad_num):
    from svi_hip.conditioning import condition_video
    ff = torch.from_numpy(synth.randn(1, 2, 3, 8, 8))
    ref = torch.from_numpy(synth.randn(2, 3, 8, 8))
    v = condition_video(ff, ref, 9, ref_pad_num)
    assert v.shape == (3, 9, 8, 8)
    assert torch.equal(v[:, :2], ff.permute(1, 0, 2, 3))
    if ref_pad_num == 0:
        assert torch.all(v[:, 2:] == 0)
    elif ref_pad_num == -1:
        assert all(torch.equal(v[:, i], ref) for i in range(2, 9))
    else:
        assert torch.equal(v[:, 2], ref) and torch.equal(v[:, 3], ref) and torch.all(v[:, 4:] == 0)


def test_oracle_shapes_and_mask():
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    ff = torch.from_numpy(np.tanh(synth.randn(3, 1, 3, 16, 16)))
    with torch.no_grad():
        y = wvo.image_condition(sd, ff, None, 5)
    assert y.shape == (1, 20, 2, 2, 2)
    assert torch.all(y[0, :4, 0] == 1) and torch.all(y[0, :4, 1] == 0)


def _golden_case(i):
    """Inputs of case i of tests/golden/image_condition.npz (tests/gen_golden.py::gen_image_condition ran the reference's
    own encode_images_adaptive on these frames): uint8 frames -> preprocess_image's float32 arithmetic."""
    name, n, cfg, pad = synth.IMAGE_CONDITION_CASES[i]
    ff = torch.from_numpy(synth.frames_to_tensor(synth.condition_frames(600 + i, n, 32, 48)))
    ref = torch.from_numpy(synth.frames_to_tensor(synth.condition_frames(650 + i, 1, 32, 48)))[0]
    return name, n, cfg, pad, ff, ref


@pytest.mark.parametrize("i", range(len(synth.IMAGE_CONDITION_CASES)))
def test_oracle_image_condition_is_pinned_to_reference(golden, i):
    """Row a22 pinned: the oracle's restatement against y as returned by SVIVideoPipeline.encode_images_adaptive (bf16)."""
    name, n, cfg, pad, ff, ref = _golden_case(i)
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    with torch.no_grad():
        y = wvo.image_condition(sd, ff, ref, 9, cfg, pad)
    want = torch.from_numpy(golden("image_condition.npz")[name])
    assert y.shape == want.shape
    assert torch.equal(y[0, :4], want[0, :4])                                   # mask: exact
    got = y.to(torch.bfloat16).float()                                          # the reference returns y in the DiT dtype
    # same fp32 arithmetic up to summation order: after the bf16 cast at most a few values sit on a rounding boundary
    diff = (got - want).abs()
    assert float(diff.max()) <= 2 ** -7 * float(want.abs().max()) and float((diff > 0).float().mean()) < 0.01, (float(diff.max()), float((diff > 0).float().mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(synth.IMAGE_CONDITION_CASES)))
def test_image_condition_matches_reference_golden(golden, i):
    import svi_hip
    name, n, cfg, pad, ff, ref = _golden_case(i)
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    v = svi_hip.WanVideoVAE.from_state_dict(sd)
    got = svi_hip.image_condition(v, ff.cuda(), ref.cuda(), 9, cfg, pad).float().cpu()      # bf16, as the reference returns it
    want = torch.from_numpy(golden("image_condition.npz")[name])
    assert got.shape == want.shape and torch.equal(got[0, :4], want[0, :4])
    diff = (got - want).abs()
    assert float(diff.max()) <= 2 ** -7 * float(want.abs().max()) and float((diff > 0).float().mean()) < 0.01, (float(diff.max()), float((diff > 0).float().mean()))


def _bf16_close(got, want):
    diff = (got - want).abs()
    return float(diff.max()) <= 2 ** -7 * float(want.abs().max()) and float((diff > 0).float().mean()) < 0.02


@pytest.mark.parametrize("i", range(len(synth.IMAGE_CONDITION_CASES)))
def test_oracle_clip_feature_is_pinned_to_reference(golden, i):
    """The other half of encode_images_adaptive's result: clip_feature = bf16(encode_image([first frame])) (svi_video.py:317, :355)."""
    from oracle import encoders_oracle as eo
    name, n, cfg, pad, ff, ref = _golden_case(i)
    sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(synth.CLIP_SEED, **synth.CLIP_TINY).items()}
    with torch.no_grad():
        got = eo.clip_encode_image(sd, ff[:1], synth.CLIP_TINY).to(torch.bfloat16).float()
    want = torch.from_numpy(golden("image_condition.npz")["clip_" + name])
    assert got.shape == want.shape and _bf16_close(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(synth.IMAGE_CONDITION_CASES)))
def test_image_condition_with_image_encoder_matches_reference_golden(golden, i):
    import svi_hip
    name, n, cfg, pad, ff, ref = _golden_case(i)
    v = svi_hip.WanVideoVAE.from_state_dict({k: torch.from_numpy(a) for k, a in synth.vae_state_dict(500).items()})
    enc = svi_hip.WanImageEncoder.from_state_dict({k: torch.from_numpy(a) for k, a in synth.clip_state_dict(synth.CLIP_SEED, **synth.CLIP_TINY).items()}, num_heads=2)
    r = svi_hip.image_condition(v, ff.cuda(), ref.cuda(), 9, cfg, pad, image_encoder=enc)
    g = golden("image_condition.npz")
    assert set(r) == {"clip_feature", "y"} and r["clip_feature"].dtype == torch.bfloat16
    assert _bf16_close(r["clip_feature"].float().cpu(), torch.from_numpy(g["clip_" + name]))
    assert torch.equal(r["y"], svi_hip.image_condition(v, ff.cuda(), ref.cuda(), 9, cfg, pad))


@pytest.mark.gpu
@pytest.mark.parametrize("n_cond,ref_pad_cfg,ref_pad_num", [(1, False, 0), (2, True, 1), (1, False, -1)])
def test_image_condition_matches_oracle(n_cond, ref_pad_cfg, ref_pad_num):
    import svi_hip
    from gpu_util import errs, report
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    v = svi_hip.WanVideoVAE.from_state_dict(sd)
    ff = torch.from_numpy(np.tanh(synth.randn(700 + n_cond, n_cond, 3, 24, 32)))
    ref = torch.from_numpy(np.tanh(synth.randn(710, 3, 24, 32)))
    with torch.no_grad():
        want = wvo.image_condition(sd, ff, ref, 9, ref_pad_cfg, ref_pad_num)
    got = svi_hip.image_condition(v, ff.cuda(), ref.cuda(), 9, ref_pad_cfg, ref_pad_num, out_dtype=torch.float32)
    r, mx, _ = errs(got, want)
    report("image_condition", n_cond=n_cond, ref_pad_num=ref_pad_num, rel_l2=r, max_abs=mx)
    assert got.shape == want.shape == (1, 20, 3, 3, 4)
    assert torch.equal(got[0, :4].cpu(), want[0, :4])          # mask: exact
    assert r < 2e-5 and mx < 2e-4, (r, mx)
