"""The pose-embedder oracle (oracle/pose_oracle.py) against the reference's own nn.Sequential + call statements (golden/pose_embed.npz)."""
import numpy as np
import torch

import synth
from conftest import rel_l2
from oracle import pose_oracle as po


def test_pose_embedder_matches_reference(golden):
    g = golden("pose_embed.npz")
    sd = {k: torch.from_numpy(v) for k, v in synth.pose_state_dict(synth.POSE_SEED).items()}
    with torch.no_grad():
        for name, shape, seed in synth.POSE_CASES:
            out = po.pose_embed(sd, torch.from_numpy(synth.pose_video(seed, *shape))).float().numpy()
            assert out.shape == g[name].shape
            assert np.array_equal(out, g[name]), (name, rel_l2(out, g[name]))        # same torch ops on the same host: bit-identical


def test_parameter_inventory_is_the_reference_sequential():
    shapes = synth.pose_param_shapes()
    assert list(shapes) == [f"{2 * i}.{leaf}" for i in range(7) for leaf in ("weight", "bias")]
    assert shapes["0.weight"] == (16, 3, 3, 3, 3) and shapes["12.weight"] == (5120, 16, 1, 2, 2)
