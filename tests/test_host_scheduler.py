"""Host-side mirror of FlowMatchScheduler against the reference's known answers (golden) — CPU only."""
import numpy as np
import torch

import synth
from svi_hip import FlowMatchScheduler, generate_noise


def test_sigmas_timesteps_and_euler_trajectory(golden):
    g = golden("flow_match.npz")
    for n in (4, 10, 50):
        s = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(n, shift=5.0)
        assert np.array_equal(s.sigmas.numpy(), g[f"sigmas_{n}"])
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps_{n}"])
        x = torch.from_numpy(synth.randn(7, 2, 3))
        v = torch.from_numpy(synth.randn(8, 2, 3))
        for i in range(n):
            x = s.step(v, s.timesteps[i], x)
            assert np.array_equal(x.numpy(), g[f"euler_traj_{n}"][i])


def test_last_step_goes_to_zero_sigma():
    s = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    s.set_timesteps(10, shift=5.0)
    assert abs(s.step_delta(s.timesteps[-1]) + float(s.sigmas[-1])) < 1e-7
    assert abs(s.step_delta(s.timesteps[0]) - float(s.sigmas[1] - s.sigmas[0])) < 1e-7


def test_default_constructor_matches_reference_defaults():
    s = FlowMatchScheduler()
    assert len(s.sigmas) == 100 and abs(float(s.sigmas[0]) - 1.0) < 1e-6


def test_add_noise_and_training_target():
    s = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    s.set_timesteps(1000, training=True)
    x, n = torch.ones(3), torch.zeros(3)
    t = s.timesteps[500]
    assert torch.allclose(s.add_noise(x, n, t), (1 - s.sigmas[500]) * x)
    assert torch.equal(s.training_target(x, n, t), n - x)
    assert s.training_weight(t) > 0


def test_generate_noise_is_cpu_generator(golden):
    g = golden("denoise_tiny.npz")
    assert np.array_equal(generate_noise((8,), seed=11, device="cpu", dtype=torch.float32).numpy(), g["noise_head"])
