"""CPU: the encoder oracle (oracle/encoders_oracle.py) against outputs of the reference's own WanTextEncoder / WanPrompter.encode_prompt
and WanImageEncoder.encode_image (tests/golden/t5_encoder.npz, clip_encoder.npz, made by tests/gen_golden.py), and the host-side bucket
table of the C library against T5RelativeEmbedding's own."""
import numpy as np
import pytest
import torch

import synth
from conftest import rel_l2
from oracle import encoders_oracle as eo


def _t(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def test_bucket_table_matches_reference(golden):
    g = golden("t5_encoder.npz")
    assert np.array_equal(eo.relative_position_buckets(32, 128, 512), g["buckets_512"])
    from svi_hip import encoders
    assert np.array_equal(np.array(encoders.relative_position_buckets(32, 128, 512), np.int32), g["buckets_512"])       # host-only C entry
    # a shorter table is the middle of a longer one
    assert encoders.relative_position_buckets(32, 128, 40) == [int(v) for v in g["buckets_512"][511 - 39:511 + 40]]


@pytest.mark.parametrize("name,L,valid,seed", synth.T5_TINY_CASES)
def test_t5_tiny_vs_reference(golden, name, L, valid, seed):
    g = golden("t5_encoder.npz")
    sd = _t(synth.t5_state_dict(synth.T5_SEED, **synth.T5_TINY))
    ids, _ = synth.t5_ids(seed, L, valid, synth.T5_TINY["vocab"])
    with torch.no_grad():
        o32 = eo.t5_encode(sd, torch.from_numpy(ids[0]), valid, synth.T5_TINY).numpy()
        o16 = eo.t5_encode(sd, torch.from_numpy(ids[0]), valid, synth.T5_TINY, "bf16").numpy()
    assert rel_l2(o32, g[f"{name}_fp32"]) < 2e-6
    # the bf16 restatement against the reference module cast to bf16: same rounding points, different summation order inside the matmuls
    assert rel_l2(o16, g[f"{name}_bf16"]) < 6e-3
    assert rel_l2(g[f"{name}_bf16"], g[f"{name}_fp32"]) < 2e-2


def test_t5_xxl_block_vs_reference(golden):
    g = golden("t5_encoder.npz")
    cfg = synth.T5_XXL_BLOCK
    name, L, valid, seed = synth.T5_XXL_CASE
    sd = _t(synth.t5_state_dict(synth.T5_SEED + 1, **cfg))
    ids, _ = synth.t5_ids(seed, L, valid, cfg["vocab"])
    with torch.no_grad():
        o32 = eo.t5_encode(sd, torch.from_numpy(ids[0]), valid, cfg).numpy()
    assert rel_l2(o32[synth.T5_XXL_ROWS], g["xxl_fp32"]) < 2e-6


@pytest.mark.parametrize("name,shape,seed", synth.CLIP_TINY_CASES)
def test_clip_tiny_vs_reference(golden, name, shape, seed):
    g = golden("clip_encoder.npz")
    sd = _t(synth.clip_state_dict(synth.CLIP_SEED, **synth.CLIP_TINY))
    with torch.no_grad():
        o = eo.clip_encode_image(sd, torch.from_numpy(synth.clip_image(seed, *shape)), synth.CLIP_TINY).numpy()
    assert o.shape == g[name].shape
    assert rel_l2(o, g[name]) < 2e-6


def test_clip_h14_block_vs_reference(golden):
    g = golden("clip_encoder.npz")
    name, shape, seed = synth.CLIP_H_CASE
    sd = _t(synth.clip_state_dict(synth.CLIP_SEED + 1, **synth.CLIP_H_BLOCK))
    with torch.no_grad():
        o = eo.clip_encode_image(sd, torch.from_numpy(synth.clip_image(seed, *shape)), synth.CLIP_H_BLOCK).numpy()
    assert rel_l2(o[:, synth.CLIP_H_ROWS], g[name]) < 2e-6


def test_t5_real_depth_vs_reference(golden):
    """24 blocks (the depth of umT5-XXL) at the tiny width: what the bf16 rounding points accumulate to."""
    g = golden("t5_encoder.npz")
    cfg = synth.T5_DEEP
    name, L, valid, seed = synth.T5_DEEP_CASE
    sd = _t(synth.t5_state_dict(synth.T5_SEED + 2, **cfg))
    ids, _ = synth.t5_ids(seed, L, valid, cfg["vocab"])
    with torch.no_grad():
        o32 = eo.t5_encode(sd, torch.from_numpy(ids[0]), valid, cfg).numpy()
        o16 = eo.t5_encode(sd, torch.from_numpy(ids[0]), valid, cfg, "bf16").numpy()
    assert rel_l2(o32, g["deep_fp32"]) < 5e-6
    # bf16 noise grows with depth: the reference's own bf16 module is 3.0e-2 from its fp32 run here; the restatement is as close to the
    # fp32 run (2.9e-2) and as far from the bf16 run as two independent bf16 evaluations are (3.0e-2 <= sqrt(2) x the gap)
    gap = rel_l2(g["deep_bf16"], g["deep_fp32"])
    assert gap < 4e-2 and rel_l2(o16, g["deep_fp32"]) < 1.2 * gap and rel_l2(o16, g["deep_bf16"]) < 1.5 * gap


def test_clip_real_depth_vs_reference(golden):
    """32 blocks, 31 used (ViT-H/14's depth) at the tiny width."""
    g = golden("clip_encoder.npz")
    name, shape, seed = synth.CLIP_DEEP_CASE
    sd = _t(synth.clip_state_dict(synth.CLIP_SEED + 2, **synth.CLIP_DEEP))
    with torch.no_grad():
        o = eo.clip_encode_image(sd, torch.from_numpy(synth.clip_image(seed, *shape)), synth.CLIP_DEEP).numpy()
    assert rel_l2(o, g[name]) < 5e-6
