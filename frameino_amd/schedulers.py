"""Samplers used by the FrameINO pipelines (host side; the per-step update itself is the fino_cfg_euler_step kernel).

`FlowMatchEulerDiscreteScheduler` mirrors the diffusers class the reference constructs
(/root/reference/pipelines/pipeline_wan_i2v_motion_FrameINO.py:179, :762, :891; training config
config/train_wan_motion_FrameINO.yaml:43-50: shift=5).  diffusers is third-party and unpinned by the reference
(requirements.txt:12): the schedule below restates its published algorithm (static shift applied in __init__ to get
sigma_min/max and again in set_timesteps; step: x += (sigma_next - sigma) * v in fp32, cast to the model dtype).
"""
import numpy as np
import torch


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, shift=1.0, **unused):
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, shift=shift)
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.sigma_min = sig[-1].item()
        self.sigma_max = sig[0].item()
        self._step_index = None
        self.num_inference_steps = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps, device=None, **unused):
        n, shift = self.config.num_train_timesteps, self.config.shift
        ts = np.linspace(self.sigma_max * n, self.sigma_min * n, num_inference_steps)
        sig = ts / n
        sig = shift * sig / (1 + (shift - 1) * sig)
        sig = torch.from_numpy(sig).to(dtype=torch.float32, device=device)
        self.timesteps = sig * n
        self.sigmas = torch.cat([sig, torch.zeros(1, device=sig.device)])
        self.num_inference_steps = num_inference_steps
        self._step_index = None
        # per-step increments, resident on the device: the kernel reads dt[i] through a pointer (graph-replayable)
        self.dts = (self.sigmas[1:] - self.sigmas[:-1]).contiguous()

    def index_for_timestep(self, timestep):
        idx = (self.timesteps == timestep).nonzero()
        return idx[1 if len(idx) > 1 else 0].item()

    def step(self, model_output, timestep, sample, return_dict=True, **unused):
        """Tensor-in/tensor-out step (API parity with diffusers); runs the fino_cfg_euler_step kernel (no CFG).
        The pipeline calls the fused CFG+Euler kernel directly instead."""
        from . import ops
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        shp = sample.shape
        lat = sample.to(torch.float32).reshape(shp[-4:]).contiguous().clone()
        pred = model_output.reshape(shp[-4:]).contiguous()
        ops.cfg_euler_step_(lat, pred, None, 1.0, self.dts[self._step_index:self._step_index + 1].to(lat.device),
                            round_out=True)
        self._step_index += 1
        prev = lat.reshape(shp).to(model_output.dtype)
        return (prev,) if not return_dict else _Cfg(prev_sample=prev)


class CogVideoXDDIMScheduler:
    """diffusers' CogVideoXDDIMScheduler as the CogVideoX-5B-I2V repo configures it (third-party, restated, unpinned):
    scaled-linear betas, SNR shift, zero-terminal-SNR rescale, trailing spacing, v-prediction, eta = 0.
    Call sites: pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:771 (retrieve_timesteps), :916 (step)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.0120, snr_shift_scale=1.0,
                 rescale_betas_zero_snr=True, set_alpha_to_one=True, **unused):
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, prediction_type="v_prediction",
                           timestep_spacing="trailing")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:
            s = ac.sqrt()
            a0, at = s[0].clone(), s[-1].clone()
            ac = ((s - at) * (a0 / (a0 - at))) ** 2
        self.alphas_cumprod = ac
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64) if set_alpha_to_one else ac[0]

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None, **unused):
        n = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        self.timesteps = torch.from_numpy(ts).to(device)
        # per-step coefficients {sa, sb, ca, cb} of  x0 = sa*x - sb*v ; x' = ca*x + cb*x0  (device-resident table)
        rows = []
        for t in ts.tolist():
            prev = t - n // num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
            ca = ((1 - a_p) / (1 - a_t)) ** 0.5
            cb = a_p ** 0.5 - a_t ** 0.5 * ca
            rows.append([float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(ca), float(cb)])
        self.coefs = torch.tensor(rows, dtype=torch.float32, device=device)
