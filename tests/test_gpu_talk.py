"""-m gpu: the talk variant — AudioProjModel + per-block audio cross-attention + the three-way-guidance sampler — against
the reference's own WanModel(enable_multitalk=True) forward and its own _sample_with_multitalk (golden/dit_tiny_talk.npz), and the oracle.
Bounds as for the plain forward: <= 2e-2 vs the reference's bf16 output, <= 1e-2 vs the oracle with bf16 rounding points; sampler <= 5e-2."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import bf16r, dev, errs, report
from oracle import wan_dit_oracle as wdo
from test_oracle_dit import make_cfg

pytestmark = pytest.mark.gpu


def _setup():
    import svi_hip
    c, seed, (f, h, w) = synth.TINY_DIT_TALK, synth.TALK_SEED, synth.TALK_GRID
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **c).items()}
    m = svi_hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    x = torch.from_numpy(synth.randn(seed + 1, 1, 16, f, 2 * h, 2 * w))
    ctx = torch.from_numpy(synth.text_context(seed + 2, 20, c["text_dim"], 13))
    kw = dict(clip_feature=torch.from_numpy(synth.randn(seed + 3, 1, 257, 1280)), y=torch.from_numpy(synth.randn(seed + 4, 1, 20, f, 2 * h, 2 * w)))
    aud = tuple(torch.from_numpy(a) for a in synth.audio_windows(seed + 5, f))
    return svi_hip, m, sd, c, x, ctx, kw, aud


def test_talk_forward_matches_reference(golden):
    hip, m, sd, c, x, ctx, kw, aud = _setup()
    g = golden("dit_tiny_talk.npz")
    ts = torch.tensor([637.5])
    kwd = {k: dev(v) for k, v in kw.items()}
    got = hip.model_fn_wan_talk_video(m, dev(x), ts, dev(ctx), audio_embed_tuple=tuple(dev(a) for a in aud), **kwd)
    sdb = {k: bf16r(v) for k, v in sd.items()}
    with torch.no_grad():
        want = wdo.dit_forward(sdb, make_cfg(c), x, ts, ctx, rounding="bf16", audio_embed_tuple=aud, **kw)
    r_or = errs(got, want)[0]
    r16, r32 = errs(got, g["out_bf16"])[0], errs(got, g["out_fp32"])[0]
    plain = m.forward(dev(x), ts, dev(ctx), **kwd)                       # model_fn_wan_talk_video disarmed the audio again
    r_plain = errs(plain, g["out_fp32_no_audio"])[0]
    report("talk_forward", vs_oracle_bf16=r_or, vs_ref_bf16=r16, vs_ref_fp32=r32, no_audio_vs_ref=r_plain)
    assert r_or < 1e-2 and r16 < 2e-2 and r_plain < 2e-2, (r_or, r16, r32, r_plain)
    assert errs(got, g["out_fp32_no_audio"])[0] > 0.1                     # the audio branch is really in the result


def test_talk_sampler_matches_reference(golden):
    hip, m, sd, c, x, ctx, kw, aud = _setup()
    g = golden("dit_tiny_talk.npz")
    seed, (f, h, w) = synth.TALK_SEED, synth.TALK_GRID
    lat = hip.generate_noise((1, 16, f, 2 * h, 2 * w), seed=31, device="cpu", dtype=torch.float32)
    null = tuple(torch.from_numpy(a) for a in synth.audio_windows(seed + 7, f))
    neg = torch.from_numpy(synth.text_context(seed + 12, 20, c["text_dim"], 4))
    out = hip.DenoiseLoop(m).sample_multitalk(dev(lat), dev(ctx), dev(neg), aud, null, num_inference_steps=3, text_scale=5.0, audio_scale=4.0,
                                             **{k: dev(v) for k, v in kw.items()})
    r = errs(out, g["sampler_latents"])[0]
    report("talk_sampler", rel_l2=r)
    assert r < 5e-2, r


def test_cfg3_step_is_the_reference_arithmetic():
    """uncond + s_t*(cond - drop) + s_a*(drop - uncond), then the Euler update: bf16 tensor ops in the reference's order, bit for bit."""
    import svi_hip
    g = torch.Generator().manual_seed(3)
    lat, c, u, d = (torch.randn((3, 1000, 7), generator=g).to(torch.bfloat16) for _ in range(4))
    st, sa, ds = 5.0, 4.0, -0.0625
    want = u + st * (c - d) + sa * (d - u)                                # svi_video_talk.py:457-459 on bf16 tensors
    want = lat + want * ds                                                # FlowMatchScheduler.step, flow_match.py:63
    got = svi_hip.cfg3_step_(lat.cuda().clone(), c.cuda(), u.cuda(), d.cuda(), st, sa, ds)
    assert torch.equal(got.cpu(), want)
