"""-m gpu: the dance variant — the pose embedder on the HIP convolution kernels against the reference's own module
(golden/pose_embed.npz) and the oracle, and the sampler's branch rule (add_condition on the conditional forward only) against the
reference's own _sample_with_dance_video (golden/dance_sampler.npz).

Embedder: fp32 convolutions, one final rounding to bf16 -> compared as bf16 values: rel-L2 <= 2e-3 and no element more than one bf16 ulp of the
reference value apart (summation order inside the convolutions decides rounding ties; measured rel-L2 7e-6)."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import dev, errs, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def embedder():
    import svi_hip
    sd = {k: torch.from_numpy(v) for k, v in synth.pose_state_dict(synth.POSE_SEED).items()}
    return svi_hip.PoseEmbedder.from_state_dict({"dwpose_embedding." + k: v for k, v in sd.items()}), sd       # checkpoint-style keys


@pytest.mark.parametrize("case", synth.POSE_CASES, ids=lambda c: c[0])
def test_pose_embedder_matches_reference(embedder, golden, case):
    m, _ = embedder
    name, shape, seed = case
    want = golden("pose_embed.npz")[name]
    got = m(torch.from_numpy(synth.pose_video(seed, *shape)).cuda())
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == want.shape
    r, mx, wmax = errs(got, want)
    g = got.float().cpu().numpy()
    off = np.abs(g - want) > np.maximum(np.abs(want) * 2.0 ** -7, 1e-6)        # more than one bf16 ulp of the reference value apart
    report("pose_embed", case=name, rel_l2=r, max_abs=mx, frac_differing=float((g != want).mean()), beyond_one_ulp=int(off.sum()))
    assert r < 2e-3 and not off.any(), (r, mx, int(off.sum()))


def test_pose_embedder_c2_geometry(embedder):
    """81 frames 832x480 -> (21, 30, 52) = 32760 token rows of width 5120: the DiT's own grid; deterministic and finite."""
    m, _ = embedder
    assert m.tokens(81, 480, 832) == (21, 30, 52)
    g = torch.Generator(device="cuda").manual_seed(5)
    pose = (torch.rand((3, 81, 480, 832), generator=g, device="cuda") * 255).floor()
    a = m(pose)
    assert tuple(a.shape) == (1, 32760, 5120) and torch.isfinite(a.float()).all()
    assert torch.equal(a, m(pose))
    # frame causality does not hold (ordinary convolutions), but locality does: a change in the last frames leaves the first token frames alone
    pose2 = pose.clone()
    pose2[:, 60:] = 0
    b = m(pose2)
    assert torch.equal(a[:, :30 * 52 * 8], b[:, :30 * 52 * 8]) and not torch.equal(a, b)


@pytest.mark.parametrize("name,wo", [("cond_only", False), ("cond_wo_pose", True)])
def test_dance_sampler_matches_reference(golden, name, wo):
    import svi_hip
    want = golden("dance_sampler.npz")[name]
    c, seed, grid = synth.TINY_DIT_I2V, 200, (2, 4, 4)
    f, h, w = grid
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    dit = svi_hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    lat = svi_hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=21, device="cpu", dtype=torch.float32)
    cond = dict(clip_feature=dev(synth.randn(seed + 3, 1, 257, 1280)), y=dev(synth.randn(seed + 4, 1, 20, f, 2 * h, 2 * w)),
                add_condition=dev(0.5 * synth.randn(seed + 9, 1, f * h * w, c["dim"])))
    out = svi_hip.DenoiseLoop(dit).sample(dev(lat), dev(synth.text_context(seed + 2, 16, c["text_dim"], 10)),
                                          dev(synth.text_context(seed + 12, 16, c["text_dim"], 4)), num_inference_steps=3, cfg_scale=5.0,
                                          sigma_shift=5.0, cond_wo_pose=wo, **cond)
    r = errs(out, want)[0]
    other = errs(out, golden("dance_sampler.npz")["cond_wo_pose" if not wo else "cond_only"])[0]
    report("dance_sampler", case=name, rel_l2=r, vs_other_branch_rule=other)
    assert r < 3e-2 and other > 2 * r, (r, other)          # and the two rules are told apart
