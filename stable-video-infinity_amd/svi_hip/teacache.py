"""TeaCache host logic for the HIP backend: when may a step reuse the previous step's block-stack residual?

Mirrors class TeaCache of the reference (pipelines/svi_video.py:23-72): the relative L1 change of t_mod between consecutive
steps, rescaled by a per-model polynomial, is accumulated; while the sum stays below `rel_l1_thresh` the 30/40 DiT blocks are
skipped and x + previous_residual is fed to the head.  First and last step always compute.  An opt-in approximation (off in
test_svi.py unless --tea_cache_l1_thresh is given); the reference's own object can be passed to model_fn_wan_video instead.
One object per CFG branch, as in the reference (tea_cache_posi / tea_cache_nega).
"""
from __future__ import annotations

import torch

COEFFICIENTS = {      # svi_video.py:35-40, fitted by the TeaCache authors per checkpoint
    "Wan2.1-T2V-1.3B": [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
    "Wan2.1-T2V-14B": [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
    "Wan2.1-I2V-14B-480P": [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01],
    "Wan2.1-I2V-14B-720P": [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02],
}


def _rescaled_drift(coefficients, previous: torch.Tensor, current: torch.Tensor) -> float:
    """The fitted polynomial (Horner form, double precision — what numpy.poly1d evaluates) of the mean absolute change of the modulation input
    relative to its previous mean magnitude; the tensor arithmetic stays in the tensors' dtype, as in the reference (svi_video.py:51)."""
    rel = ((current - previous).abs().mean() / previous.abs().mean()).cpu().item()
    acc = 0.0
    for c in coefficients:
        acc = acc * rel + c
    return acc


class TeaCache:
    """A drift budget: every step adds its rescaled drift to a running sum; the step is skipped while the sum stays under `rel_l1_thresh`, and a computed
    step empties the sum.  The first and the last step of a clip always compute.  Attribute names are the reference's (its own object and this one are
    interchangeable in model_fn_wan_video): step, accumulated_rel_l1_distance, previous_modulated_input, previous_residual, previous_hidden_states."""

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

    def _must_compute(self, current: torch.Tensor) -> bool:
        if self.step in (0, self.num_inference_steps - 1):
            return True
        self.accumulated_rel_l1_distance += _rescaled_drift(self.coefficients, self.previous_modulated_input, current)
        return not (self.accumulated_rel_l1_distance < self.rel_l1_thresh)

    def check(self, dit, x, t_mod) -> bool:
        """True = skip the blocks this step (the decision of svi_video.py:43-62).  `x` is not kept: the HIP forward forms the residual on the device."""
        current = t_mod.clone()
        compute = self._must_compute(current)
        if compute:
            self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = current
        self.step = (self.step + 1) % self.num_inference_steps
        return not compute
