"""Oracle restatement of the samplers the FrameINO pipelines call.  Test infrastructure.

These classes live in `diffusers` (third-party; NOT under /root/reference; unpinned by
requirements.txt:12) -- restated from the published algorithm; parity for them is UNPINNED.
Call sites: pipelines/pipeline_wan_i2v_motion_FrameINO.py:762 (set_timesteps), :891 (step);
training configures FlowMatchEuler with shift=5 (config/train_wan_motion_FrameINO.yaml:43-50).
"""
import numpy as np
import torch


class FlowMatchEulerOracle:
    """diffusers FlowMatchEulerDiscreteScheduler, static shift.  Note the published quirk that the
    shift is applied both in __init__ (to derive sigma_min/max) and again in set_timesteps."""

    order = 1

    def __init__(self, num_train_timesteps=1000, shift=5.0):
        self.n_train = num_train_timesteps
        self.shift = shift
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min = sig[-1].item()
        self.sigma_max = sig[0].item()

    def set_timesteps(self, num_inference_steps):
        ts = np.linspace(self.sigma_max * self.n_train, self.sigma_min * self.n_train, num_inference_steps)
        sig = ts / self.n_train
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        sig = torch.from_numpy(sig).to(torch.float32)
        self.timesteps = sig * self.n_train
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self.step_index = 0

    def step(self, model_output, sample):
        s, sn = self.sigmas[self.step_index], self.sigmas[self.step_index + 1]
        prev = sample.to(torch.float32) + (sn - s) * model_output
        self.step_index += 1
        return prev.to(model_output.dtype)     # diffusers casts back to the model-output dtype
