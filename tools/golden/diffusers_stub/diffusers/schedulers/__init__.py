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


class CogVideoXDPMScheduler(CogVideoXDDIMScheduler):
    """Stand-in for diffusers.CogVideoXDPMScheduler (third-party, restated: the arithmetic is the builder's ONE
    restatement, /root/repo/oracle/schedulers.py::CogDPMOracle -- get_variables / get_mult / step of the published
    SDE-DPM-Solver++(2M) form; UNPINNED).  Interface of the reference's call site
    (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:915-926)."""

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta=0.0,
             use_clipped_model_output=False, generator=None, variance_noise=None, return_dict=False):
        import os
        import sys
        repo = os.path.abspath(os.path.join(os.path.dirname(__file__), *[".."] * 5))
        if repo not in sys.path:
            sys.path.append(repo)
        from oracle.schedulers import CogDPMOracle
        o = getattr(self, "_oracle", None)
        if o is None or o.n != self.num_inference_steps:
            o = self._oracle = CogDPMOracle(self.config.num_train_timesteps, self.config.beta_start, self.config.beta_end,
                                            self.config.snr_shift_scale)
            o.set_timesteps(self.num_inference_steps)
        tb = None if timestep_back is None else int(timestep_back)
        return o.step(model_output, old_pred_original_sample, int(timestep), tb, sample, generator=generator)


class UniPCMultistepScheduler(ConfigMixin):
    """Stand-in for diffusers.UniPCMultistepScheduler as Wan-AI/Wan2.2-TI2V-5B-Diffusers configures it (third-party,
    restated tensor-op by tensor-op from the published algorithm: flow sigmas, flow_prediction, predict_x0, bh2,
    solver_order 2, lower_order_final, final sigma 0).  Lets the REFERENCE pipeline's own loop drive a UniPC run."""
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type="flow_prediction", flow_shift=5.0,
                 use_flow_sigmas=True, solver_type="bh2", predict_x0=True, lower_order_final=True,
                 final_sigmas_type="zero"):
        assert use_flow_sigmas and prediction_type == "flow_prediction" and predict_x0 and solver_type == "bh2"
        self.model_outputs = [None] * solver_order
        self._step_index = None

    def set_timesteps(self, num_inference_steps, device=None):
        n, shift = self.config.num_train_timesteps, self.config.flow_shift
        alphas = np.linspace(1, 1 / n, num_inference_steps + 1)
        sig = 1.0 - alphas
        sig = np.flip(shift * sig / (1 + (shift - 1) * sig))[:-1].copy()
        self.timesteps = torch.from_numpy((sig * n).copy()).to(device=device, dtype=torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.num_inference_steps = num_inference_steps
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None
        self.this_order = 0

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _update(self, x, m0, others, s_t, s_0, rks_lams, model_t=None):
        """shared body of multistep_uni_p_bh_update (model_t None) / multistep_uni_c_bh_update"""
        alpha_t = 1 - s_t
        h = self._lam(s_t) - self._lam(s_0)
        rks, d1s = [], []
        for mi, lam_i in zip(others, rks_lams):
            rk = (lam_i - self._lam(s_0)) / h
            rks.append(rk)
            d1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack(rks)
        order = len(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.stack(b)
        x_t_ = s_t / s_0 * x - alpha_t * h_phi_1 * m0
        if model_t is None:                                            # predictor
            if d1s:
                rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
                pred = sum(rhos_p[k] * d1s[k] for k in range(len(d1s)))
            else:
                pred = 0
            return (x_t_ - alpha_t * b_h * pred).to(x.dtype)
        rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        corr = sum(rhos_c[k] * d1s[k] for k in range(len(d1s))) if d1s else 0
        return (x_t_ - alpha_t * b_h * (corr + rhos_c[-1] * (model_t - m0))).to(x.dtype)

    def step(self, model_output, timestep, sample, return_dict=True, **kw):
        if self._step_index is None:
            self._step_index = (self.timesteps == timestep).nonzero()[0].item()
        i = self._step_index
        use_corrector = i > 0 and self.last_sample is not None
        sigma = self.sigmas[i]
        m_t = sample - sigma * model_output                            # convert_model_output: flow prediction -> x0
        if use_corrector:
            o = self.this_order
            others = [self.model_outputs[-(k + 1)] for k in range(1, o)]
            lams = [self._lam(self.sigmas[i - (k + 1)]) for k in range(1, o)]
            sample = self._update(self.last_sample, self.model_outputs[-1], others, self.sigmas[i], self.sigmas[i - 1],
                                  lams, model_t=m_t)
        self.model_outputs = self.model_outputs[1:] + [m_t]
        this_order = min(self.config.solver_order, len(self.timesteps) - i) if self.config.lower_order_final \
            else self.config.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        others = [self.model_outputs[-(k + 1)] for k in range(1, self.this_order)]
        lams = [self._lam(self.sigmas[i - k]) for k in range(1, self.this_order)]
        prev = self._update(sample, self.model_outputs[-1], others, self.sigmas[i + 1], self.sigmas[i], lams)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,)
