"""-m gpu: value checks at the BASELINE configurations' own sizes, against outputs of the REFERENCE itself
(tests/golden/c1_e2e.npz, vae_c2.npz, dit_block_14b.npz — made by tests/gen_golden.py from /root/reference) and, where the
host can afford it, the CPU oracle on the same box.

  C1  (BASELINE configs[0]): full 30-layer Wan2.1-T2V-1.3B, 17 frames 256x256, 10 flow-match steps, CFG 5, then VAE decode.
      latents rel-L2 <= 5e-2 against the reference's fp32 run; the decoded video of the reference's latents
      rel-L2 <= 2e-5 (VAE alone at the C1 size); the end-to-end video rel-L2 <= 5e-2 (latent error through a random decoder).
      Measured (profiles/r2a_parity_report.jsonl): latents 2.34e-2 vs the reference's fp32 run — the reference's own bf16 run differs
      from it by 2.33e-2 — and 1.29e-2 vs its bf16 run; video 2.5e-2; VAE on the reference's latents 4.0e-6.
  C2  (configs[1]) VAE size: 2 latent frames at 60x104 -> 5 frames 480x832 and back, rel-L2 <= 2e-5, max-abs <= 2e-4; the
      headline 21-frame decode is tied to it by frame causality (its first 5 frames are that decode).
  C4  (configs[3]) widths: one DiTBlock at dim 5120 / 40 heads / ffn 13824 with the 257-token image branch,
      rel-L2 <= 6e-3 against the oracle with bf16 rounding points, <= 1.5e-2 against the reference's fp32 rows.
"""
import numpy as np
import pytest
import torch

import synth
from conftest import rel_l2
from gpu_util import bf16r, dev, errs, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import svi_hip
    return svi_hip


@pytest.fixture(scope="module")
def vae(hip):
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}
    return hip.WanVideoVAE.from_state_dict(sd), sd


# ------------------------------------------------------------------------------------------------------------------ C1
def test_c1_end_to_end_vs_reference(hip, vae, golden):
    g = golden("c1_e2e.npz")
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **cfg).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(cfg), **cfg)
    del sd
    noise = hip.generate_noise((1, 16, 5, 32, 32), seed=0, device="cpu", dtype=torch.float32)
    pos = torch.from_numpy(synth.text_context(seed + 1, 512, cfg["text_dim"], 64))
    neg = torch.from_numpy(synth.text_context(seed + 2, 512, cfg["text_dim"], 64))
    lat = hip.DenoiseLoop(m).sample(dev(noise), dev(pos), dev(neg), num_inference_steps=10, cfg_scale=5.0, sigma_shift=5.0)
    r32 = errs(lat[0], g["latents_fp32"])[0]
    r16 = errs(lat[0], g["latents_bf16"])[0]
    ref_gap = rel_l2(g["latents_bf16"], g["latents_fp32"])
    v, _ = vae
    k = synth.C1_VIDEO_STRIDE
    video_ref_lat = v.decode([torch.from_numpy(g["latents_fp32"]).cuda()], device="cuda")[0]
    assert tuple(video_ref_lat.shape) == tuple(int(a) for a in g["video_shape"])
    rv, mxv, _ = errs(video_ref_lat[:, :, ::k, ::k], g["video_sample"])
    video = v.decode([lat[0].float()], device="cuda")[0]
    re2e, mxe, _ = errs(video[:, :, ::k, ::k], g["video_sample"])
    report("c1_e2e", latents_vs_ref_fp32=r32, latents_vs_ref_bf16=r16, ref_bf16_vs_fp32=ref_gap, vae_on_ref_latents=rv,
           vae_on_ref_latents_maxabs=mxv, video_e2e=re2e, video_e2e_maxabs=mxe)
    assert r32 < 5e-2, (r32, r16, ref_gap)
    assert rv < 2e-5 and mxv < 2e-4, (rv, mxv)
    assert re2e < 5e-2, (re2e, mxe)


def test_c1_fifty_steps_vs_reference(hip, golden):
    """The headline's HORIZON: fifty CFG-5 flow-match steps — the loop of svi_video.py:392-421 with flow_match.py:53-64's update —
    on the C1 grid (1280 tokens, the 30-layer 1.3B architecture), against the reference's own fifty steps in fp32 and, as the yardstick, the reference's own
    bf16 run of the same loop (golden/c1_50step.npz: latents after steps 1, 5, 10, 20, 30, 40, 50).  Stated bound at every kept step:
        HIP vs the reference's fp32 latents <= max(5e-2, 1.5 x the reference's own bf16-vs-fp32 gap at that step)
    and the hipGraph-replayed loop gives the eager loop's bits."""
    g = golden("c1_50step.npz")
    keep = [int(k) for k in g["steps"]]
    assert keep == list(synth.C1_50_KEEP)
    cfg, seed = synth.WAN_1_3B, synth.C1_SEED
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **cfg).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(cfg), **cfg)
    del sd
    noise = hip.generate_noise((1, 16, 5, 32, 32), seed=0, device="cpu", dtype=torch.float32)
    pos = dev(torch.from_numpy(synth.text_context(seed + 1, 512, cfg["text_dim"], 64)))
    neg = dev(torch.from_numpy(synth.text_context(seed + 2, 512, cfg["text_dim"], 64)))
    ref16 = synth.bf16_from_bits(g["latents_bf16_bits"])

    def run(graph):
        loop = hip.DenoiseLoop(m, graph=graph)
        loop.scheduler.set_timesteps(50, shift=5.0)
        lat = dev(noise).to(torch.bfloat16).contiguous().clone()
        ts = loop.scheduler.timesteps.to("cuda", torch.float32)
        kept = []
        m.context_cache(True)
        try:
            for i, t in enumerate(loop.scheduler.timesteps):
                loop.step(lat, ts[i:i + 1], loop.scheduler.step_delta(t), pos, neg, 5.0)
                if i + 1 in keep:
                    kept.append(lat[0].clone())
        finally:
            loop.drop_graph()
            m.context_cache(False)
        return kept
    eager, graphed = run(False), run(True)
    rows = []
    for j, step in enumerate(keep):
        assert torch.equal(eager[j], graphed[j]), step                       # 49 replays of one captured step == 50 eager steps, bit for bit
        r32, r16 = errs(graphed[j], g["latents_fp32"][j])[0], errs(graphed[j], ref16[j])[0]
        gap = float(g["ref_gap"][j])
        assert abs(rel_l2(ref16[j], g["latents_fp32"][j]) - gap) < 1e-6
        rows.append((step, r32, r16, gap))
        assert r32 <= max(5e-2, 1.5 * gap), (step, r32, gap)
    assert torch.isfinite(graphed[-1].float()).all()
    report("c1_50step", **{f"step_{s}": {"hip_vs_ref_fp32": a, "hip_vs_ref_bf16": b, "ref_bf16_vs_fp32": c} for s, a, b, c in rows})


# ------------------------------------------------------------------------------------------------------------------ C2 VAE
def test_vae_c2_size_vs_reference(vae, golden):
    g = golden("vae_c2.npz")
    v, _ = vae
    k = synth.C2_VIDEO_STRIDE
    z = torch.from_numpy(synth.randn(511, 16, 2, 60, 104)).cuda()
    video = v.decode([z], device="cuda")[0]
    assert tuple(video.shape) == (3, 5, 480, 832)
    r, mx, _ = errs(video[:, :, ::k, ::k], g["decode_sample"])
    vid = torch.from_numpy(np.tanh(synth.randn(512, 3, 5, 480, 832))).cuda()
    lat = v.encode([vid], device="cuda")[0]
    re, mxe, _ = errs(lat, g["encode"])
    # the headline decode (21 latent frames, 12.4 GB of resident activations, multi-row pixel tiles, the 32-bit window guards):
    # its first two latent frames are `z`, so by frame causality its first 5 frames are the decode above
    z21 = torch.cat([z, torch.from_numpy(synth.randn(513, 16, 19, 60, 104)).cuda()], dim=1)
    full = v.decode([z21], device="cuda")[0]
    assert tuple(full.shape) == (3, 81, 480, 832) and torch.isfinite(full).all()
    causal = float((full[:, :5] - video).abs().max())
    rf, mxf, _ = errs(full[:, :5, ::k, ::k], g["decode_sample"])
    # ... and frames 41-45 and 77-81 of that 81-frame decode against the reference's own decode of the same 21 latent frames (golden/vae_c2_full.npz:
    # the temporal cache chain of all 20 later latent frames, not only causality of the first two — VERDICT r4 weak #1)
    g2 = golden("vae_c2_full.npz")
    r_mid, mx_mid, _ = errs(full[:, 40:45, ::k, ::k], g2["mid"])
    r_tail, mx_tail, _ = errs(full[:, 76:81, ::k, ::k], g2["tail"])
    report("vae_c2_full_81_frames", mid_rel=r_mid, mid_maxabs=mx_mid, tail_rel=r_tail, tail_maxabs=mx_tail)
    assert r_mid < 2e-5 and mx_mid < 2e-4 and r_tail < 2e-5 and mx_tail < 2e-4, (r_mid, mx_mid, r_tail, mx_tail)
    # and the same for the encoder: 81 frames whose first 5 are `vid`
    vid81 = torch.cat([vid, torch.from_numpy(np.tanh(synth.randn(514, 3, 76, 480, 832))).cuda()], dim=1)
    lat21 = v.encode([vid81], device="cuda")[0]
    assert tuple(lat21.shape) == (16, 21, 60, 104) and torch.isfinite(lat21).all()
    causal_e = float((lat21[:, :2] - lat).abs().max())
    rfe, _, _ = errs(lat21[:, :2], g["encode"])
    report("vae_c2", decode_rel=r, decode_maxabs=mx, encode_rel=re, encode_maxabs=mxe, full_decode_first5_rel=rf,
           full_decode_causality_maxabs=causal, full_encode_first2_rel=rfe, full_encode_causality_maxabs=causal_e)
    assert r < 2e-5 and mx < 2e-4, (r, mx)
    assert re < 2e-5 and mxe < 2e-4, (re, mxe)
    assert rf < 2e-5 and mxf < 2e-4 and causal < 1e-5, (rf, mxf, causal)
    assert rfe < 2e-5 and causal_e < 1e-5, (rfe, causal_e)


def test_vae_c2_default_tiling_runs_and_agrees_in_the_interior(vae):
    """tiled=True with the pipelines' default tiles (30,52)/(15,26) at the C2 latent size: 9 tiles.  Pinned to the reference at small
    sizes (test_gpu_vae.py); here: shape, finiteness, clamp, determinism, and corner pixels far from every seam equal the corner
    tile's own decode."""
    v, _ = vae
    z = torch.from_numpy(synth.randn(515, 16, 1, 60, 104)).cuda()
    a = v.decode([z], device="cuda", tiled=True, tile_size=(30, 52), tile_stride=(15, 26))[0]
    b = v.decode([z], device="cuda", tiled=True, tile_size=(30, 52), tile_stride=(15, 26))[0]
    assert tuple(a.shape) == (3, 1, 480, 832) and torch.isfinite(a).all() and float(a.abs().max()) <= 1.0
    assert torch.equal(a, b)
    corner = v.decode([z[:, :, :30, :52].contiguous()], device="cuda")[0]
    assert torch.equal(a[:, :, :120, :208], corner[:, :, :120, :208])          # weight 1, mask 1: value / 1 exactly


# ------------------------------------------------------------------------------------------------------------------ C4 widths
def test_block_at_14b_i2v_widths(hip, golden):
    from oracle import wan_dit_oracle as wdo
    from test_oracle_dit import make_cfg
    g = golden("dit_block_14b.npz")
    cfg = dict(synth.WAN_14B_I2V, num_layers=1)
    seed, grid, nt = synth.B14_SEED, synth.B14_GRID, 512
    f, h, w = grid
    L = f * h * w
    sd = {k: torch.from_numpy(v) for k, v in synth.dit_state_dict(seed, **cfg).items()}
    m = hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=40, **cfg)
    bx = torch.from_numpy(synth.randn(seed + 5, 1, L, cfg["dim"]))
    bctx = torch.from_numpy(synth.randn(seed + 6, 1, nt + 257, cfg["dim"]))
    btm = torch.from_numpy(0.5 * synth.randn(seed + 7, 1, 6, cfg["dim"]))
    got = m.block_forward(0, dev(bx), dev(bctx), dev(btm), grid)
    rows = [int(r) for r in g["rows"]]
    assert rows == synth.B14_ROWS(L)
    r32 = errs(got[0, rows], g["block_fp32"])[0]
    r16 = errs(got[0, rows], g["block_bf16"])[0]
    sdb = {k: bf16r(v) for k, v in sd.items() if k.startswith("blocks.0.")}
    with torch.no_grad():
        want = wdo.dit_block(sdb, "blocks.0.", bf16r(bx), bf16r(bctx), bf16r(btm), wdo.rope_table_3d(128, grid), make_cfg(cfg), "bf16")
    r_or, mx, _ = errs(got, want)
    report("dit_block_14b", vs_oracle_bf16=r_or, vs_ref_fp32=r32, vs_ref_bf16=r16, ref_bf16_vs_fp32=rel_l2(g["block_bf16"], g["block_fp32"]),
           max_abs=mx)
    assert r_or < 6e-3 and r32 < 1.5e-2, (r_or, r32, r16)
