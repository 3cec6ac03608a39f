"""CPU oracle for the flow-matching schedule, the Euler step and the CFG denoise loop —
TEST INFRASTRUCTURE ONLY (see oracle/wan_dit_oracle.py header for the import rule).

Parity status: PINNED by known-answer sigmas/timesteps produced by the reference
(`FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)`), stored in
tests/golden/flow_match.npz by tests/gen_golden.py, and by SURVEY.md §8c's literal values.

Restates: diffsynth/schedulers/flow_match.py:31-64 (set_timesteps, step),
          diffsynth/pipelines/svi_video.py:392-421 (_sample_with_regular_video),
          diffsynth/pipelines/base.py:140-143 (generate_noise).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch


def shifted_sigmas(num_steps: int, shift: float = 5.0, sigma_max: float = 1.0, sigma_min: float = 0.0,
                   denoising_strength: float = 1.0, extra_one_step: bool = True) -> np.ndarray:
    """fp32 sigma ladder: linspace then sigma <- shift*sigma / (1+(shift-1)*sigma)."""
    start = sigma_min + (sigma_max - sigma_min) * denoising_strength
    if extra_one_step:
        lin = torch.linspace(start, sigma_min, num_steps + 1)[:-1]
    else:
        lin = torch.linspace(start, sigma_min, num_steps)
    sig = shift * lin / (1 + (shift - 1) * lin)
    return sig.numpy().astype(np.float32)


def timesteps_from_sigmas(sigmas: np.ndarray, num_train_timesteps: int = 1000) -> np.ndarray:
    return (torch.from_numpy(sigmas) * num_train_timesteps).numpy()


def euler_delta(sigmas: np.ndarray, step_index: int) -> float:
    """sigma_next - sigma, with sigma_next = 0 after the last step (flow_match.py:57-63)."""
    s = torch.from_numpy(sigmas)
    nxt = torch.tensor(0.0) if step_index + 1 >= len(sigmas) else s[step_index + 1]
    return float(nxt - s[step_index])


def seeded_noise(shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """fp32 noise from torch's CPU generator — the only source that reproduces the reference's latents."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cpu", dtype=torch.float32)


def denoise_loop(model: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                 latents: torch.Tensor, ctx_pos: torch.Tensor, ctx_neg: torch.Tensor,
                 num_steps: int, cfg_scale: float = 5.0, shift: float = 5.0,
                 rounding: Optional[str] = None) -> torch.Tensor:
    """50x{cond, uncond, combine, Euler}.  `model(latents, timestep[1], context) -> velocity`."""
    rnd = (lambda t: t) if rounding is None else (lambda t: t.to(torch.bfloat16).to(torch.float32))
    sig = shifted_sigmas(num_steps, shift)
    ts = timesteps_from_sigmas(sig)
    latents = rnd(latents)
    for i in range(num_steps):
        t = torch.tensor([ts[i]], dtype=torch.float32)
        if cfg_scale != 1.0:
            cond = model(latents, t, ctx_pos)
            unc = model(latents, t, ctx_neg)
            v = rnd(unc + rnd(cfg_scale * rnd(cond - unc)))
        else:
            v = model(latents, t, ctx_pos)
        latents = rnd(latents + rnd(v * euler_delta(sig, i)))
    return latents
