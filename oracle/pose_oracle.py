"""CPU oracle for the dance variant's pose embedder — TEST INFRASTRUCTURE ONLY (see oracle/wan_dit_oracle.py header for the import rule).

Parity status: PINNED.  tests/golden/pose_embed.npz holds the output of the reference's own `dwpose_embedding` nn.Sequential (the
expression at pipelines/svi_video_dance.py:255-269, evaluated out of the source file by tests/gen_golden.py) driven by the reference's own
statements at :527-530; tests/test_oracle_pose.py checks this restatement against it.

Restates: svi_video_dance.py:255-269 (seven Conv3d, SiLU between them), :527-530 (first frame repeated three more times, / 255, bf16,
'b c f h w -> b (f h w) c').
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# (kernel, stride, padding) of the seven convolutions, dance:256-269
LAYERS = [((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (1, 1, 1), (1, 1, 1)),
          ((3, 3, 3), (1, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
          ((1, 2, 2), (1, 2, 2), (0, 0, 0))]


def pose_embed(sd: Dict[str, Tensor], humanpose_data: Tensor) -> Tensor:
    """humanpose_data [3, F, H, W] (0..255) -> add_condition bf16 [1, f*h*w, dim]."""
    x = humanpose_data.float().unsqueeze(0)                                            # :528
    x = torch.cat([x[:, :, :1].repeat(1, 1, 3, 1, 1), x], dim=2) / 255.0               # :529
    for i, (_, stride, pad) in enumerate(LAYERS):
        x = F.conv3d(x, sd[f"{2 * i}.weight"], sd[f"{2 * i}.bias"], stride=stride, padding=pad)
        if i != len(LAYERS) - 1:
            x = F.silu(x)
    x = x.to(torch.bfloat16)                                                           # :529 .to(torch.bfloat16)
    b, c, f, h, w = x.shape
    return x.permute(0, 2, 3, 4, 1).reshape(b, f * h * w, c).contiguous()              # :530
