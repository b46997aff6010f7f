"""Stand-in for diffusers.schedulers (third-party, restated from published semantics)."""
import numpy as np
import torch

from ..configuration_utils import ConfigMixin, register_to_config


class FlowMatchEulerDiscreteScheduler(ConfigMixin):
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, base_shift=0.5,
                 max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096, invert_sigmas=False,
                 shift_terminal=None, use_karras_sigmas=False, use_exponential_sigmas=False,
                 use_beta_sigmas=False, time_shift_type="exponential", stochastic_sampling=False):
        timesteps = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32)
        sigmas = timesteps / num_train_timesteps
        if not use_dynamic_shifting:
            sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = sigmas * num_train_timesteps
        self._step_index = None
        self._begin_index = None
        self._shift = shift
        self.sigmas = sigmas.to("cpu")
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    @property
    def step_index(self):
        return self._step_index

    def _sigma_to_t(self, sigma):
        return sigma * self.config.num_train_timesteps

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(self._sigma_to_t(self.sigma_max), self._sigma_to_t(self.sigma_min), num_inference_steps)
        sigmas = ts / self.config.num_train_timesteps
        sigmas = self._shift * sigmas / (1 + (self._shift - 1) * sigmas)
        sigmas = torch.from_numpy(sigmas).to(dtype=torch.float32, device=device)
        self.timesteps = sigmas * self.config.num_train_timesteps
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        if schedule_timesteps is None:
            schedule_timesteps = self.timesteps
        indices = (schedule_timesteps == timestep).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()

    def step(self, model_output, timestep, sample, return_dict=True, generator=None, **kw):
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        sample = sample.to(torch.float32)
        sigma = self.sigmas[self._step_index]
        sigma_next = self.sigmas[self._step_index + 1]
        dt = sigma_next - sigma
        prev_sample = sample + dt * model_output
        self._step_index += 1
        prev_sample = prev_sample.to(model_output.dtype)
        return (prev_sample,)


def _rescale_zero_terminal_snr(alphas_cumprod):
    s = alphas_cumprod.sqrt()
    a0, at = s[0].clone(), s[-1].clone()
    s = (s - at) * (a0 / (a0 - at))
    return s ** 2


class CogVideoXDDIMScheduler(ConfigMixin):
    """Stand-in for diffusers.CogVideoXDDIMScheduler (third-party, restated: scaled-linear betas, SNR shift,
    zero-terminal-SNR rescale, trailing spacing, v-prediction; step as published)."""
    order = 1
    init_noise_sigma = 1.0

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.0120, beta_schedule="scaled_linear",
                 clip_sample=False, set_alpha_to_one=True, steps_offset=0, prediction_type="v_prediction",
                 timestep_spacing="trailing", rescale_betas_zero_snr=True, snr_shift_scale=1.0):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:
            ac = _rescale_zero_terminal_snr(ac)
        self.alphas_cumprod = ac
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else ac[0]

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, return_dict=True, **kw):
        n = self.config.num_train_timesteps
        prev_t = timestep - n // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        pred_x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output            # v-prediction
        ca = ((1 - a_prev) / (1 - a_t)) ** 0.5
        cb = a_prev ** 0.5 - a_t ** 0.5 * ca
        prev = ca * sample + cb * pred_x0
        return (prev, pred_x0)
