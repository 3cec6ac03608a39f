"""TeaCache host logic for the HIP backend: when may a step reuse the previous step's block-stack residual?

Mirrors class TeaCache of the reference (pipelines/svi_video.py:23-72): the relative L1 change of t_mod between consecutive
steps, rescaled by a per-model polynomial, is accumulated; while the sum stays below `rel_l1_thresh` the 30/40 DiT blocks are
skipped and x + previous_residual is fed to the head.  First and last step always compute.  An opt-in approximation (off in
test_svi.py unless --tea_cache_l1_thresh is given); the reference's own object can be passed to model_fn_wan_video instead.
One object per CFG branch, as in the reference (tea_cache_posi / tea_cache_nega).
"""
from __future__ import annotations

import numpy as np
import torch

COEFFICIENTS = {      # svi_video.py:35-40, fitted by the TeaCache authors per checkpoint
    "Wan2.1-T2V-1.3B": [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
    "Wan2.1-T2V-14B": [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
    "Wan2.1-I2V-14B-480P": [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01],
    "Wan2.1-I2V-14B-720P": [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02],
}


class TeaCache:
    def __init__(self, num_inference_steps: int, rel_l1_thresh: float, model_id: str):
        if model_id not in COEFFICIENTS:
            raise ValueError(f"{model_id} is not a supported TeaCache model id. Please choose a valid model id in ({', '.join(COEFFICIENTS)}).")
        self.num_inference_steps = num_inference_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.coefficients = COEFFICIENTS[model_id]
        self.step = 0
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.previous_hidden_states = None

    def check(self, dit, x, t_mod) -> bool:
        """True = skip the blocks this step (svi_video.py:43-62; the arithmetic runs in t_mod's dtype, as there)."""
        modulated_inp = t_mod.clone()
        if self.step == 0 or self.step == self.num_inference_steps - 1:
            should_calc = True
            self.accumulated_rel_l1_distance = 0
        else:
            rel = ((modulated_inp - self.previous_modulated_input).abs().mean() / self.previous_modulated_input.abs().mean()).cpu().item()
            self.accumulated_rel_l1_distance += np.poly1d(self.coefficients)(rel)
            should_calc = not (self.accumulated_rel_l1_distance < self.rel_l1_thresh)
            if should_calc:
                self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = modulated_inp
        self.step = (self.step + 1) % self.num_inference_steps
        return not should_calc
