"""FlowMatchScheduler with the reference's constructor, attributes and methods
(diffsynth/schedulers/flow_match.py:3-97): host-side scalar logic only; the tensor update of `step`
runs in the fused HIP kernel when the pipeline drives it (see pipeline.py), and as plain tensor
arithmetic when called directly like the reference."""
from __future__ import annotations

import torch


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.sigma_max, self.sigma_min = sigma_max, sigma_min
        self.inverse_timesteps, self.extra_one_step, self.reverse_sigmas = inverse_timesteps, extra_one_step, reverse_sigmas
        self.set_timesteps(num_inference_steps)

    def _ladder(self, n, denoising_strength, shift):
        start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        count = n + 1 if self.extra_one_step else n
        sig = torch.linspace(start, self.sigma_min, count)
        if self.extra_one_step:
            sig = sig[:-1]
        if self.inverse_timesteps:
            sig = torch.flip(sig, dims=[0])
        sig = shift * sig / (1 + (shift - 1) * sig)
        if self.reverse_sigmas:
            sig = 1 - sig
        return sig

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False, shift=None):
        if shift is not None:
            self.shift = shift
        self.sigmas = self._ladder(num_inference_steps, denoising_strength, self.shift)
        self.timesteps = self.sigmas * self.num_train_timesteps
        if training:
            t = self.timesteps
            bell = torch.exp(-2 * ((t - num_inference_steps / 2) / num_inference_steps) ** 2)
            bell = bell - bell.min()
            self.linear_timesteps_weights = bell * (num_inference_steps / bell.sum())

    def _index(self, timestep):
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.cpu()
        return int(torch.argmin((self.timesteps - timestep).abs()))

    def step_delta(self, timestep, to_final=False, **kwargs) -> float:
        """sigma_next - sigma for the step that starts at `timestep` (what `step` multiplies by)."""
        i = self._index(timestep)
        sigma = self.sigmas[i]
        if to_final or i + 1 >= len(self.timesteps):
            nxt = 1 if (self.inverse_timesteps or self.reverse_sigmas or kwargs.get("self_corr", False)) else 0
        else:
            nxt = self.sigmas[i + 1]
        return float(nxt - sigma)

    def step(self, model_output, timestep, sample, to_final=False, **kwargs):
        return sample + model_output * self.step_delta(timestep, to_final=to_final, **kwargs)

    def return_to_timestep(self, timestep, sample, sample_stablized):
        return (sample - sample_stablized) / self.sigmas[self._index(timestep)]

    def add_noise(self, original_samples, noise, timestep):
        s = self.sigmas[self._index(timestep)]
        return (1 - s) * original_samples + s * noise

    # interface only: the object stands in for the reference's FlowMatchScheduler wherever a pipeline holds one (flow_match.py:84-97 has these
    # two); training is outside the path this package serves (out of scope) and nothing here calls them
    def training_target(self, sample, noise, timestep):
        return noise - sample

    def training_weight(self, timestep):
        i = int(torch.argmin((self.timesteps - timestep.to(self.timesteps.device)).abs()))
        return self.linear_timesteps_weights[i]
