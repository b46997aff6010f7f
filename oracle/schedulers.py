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


class UniPCOracle:
    """Direct (tensor-op by tensor-op) restatement of diffusers' UniPCMultistepScheduler.step for the Wan2.2
    configuration (flow sigmas, flow_prediction, predict_x0, bh2, solver_order 2, lower_order_final, final sigma 0).
    Third-party, from the published algorithm: unpinned.  Used to cross-check the coefficient-folded kernel form."""

    def __init__(self, num_train_timesteps=1000, flow_shift=5.0, solver_order=2):
        self.n_train, self.shift, self.order_max = num_train_timesteps, flow_shift, solver_order

    def set_timesteps(self, n):
        alphas = np.linspace(1, 1 / self.n_train, n + 1)
        sig = 1.0 - alphas
        sig = np.flip(self.shift * sig / (1 + (self.shift - 1) * sig))[:-1].copy()
        self.timesteps = torch.from_numpy((sig * self.n_train).copy()).to(torch.int64)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.n = n
        self.model_outputs = [None] * self.order_max
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = 0

    @staticmethod
    def _als(sigma):
        return 1 - sigma, sigma

    def _lam(self, sigma):
        a, s = self._als(sigma)
        return torch.log(a) - torch.log(s)

    def _rb(self, rks, hh, order):
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / b_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return h_phi_1, b_h, torch.stack(R), torch.stack(b)

    def _predict(self, sample, order):
        m0, x = self.model_outputs[-1], sample
        st, s0 = self.sigmas[self.step_index + 1], self.sigmas[self.step_index]
        at, _ = self._als(st)
        h = self._lam(st) - self._lam(s0)
        rks, d1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lam(self.sigmas[self.step_index - i]) - self._lam(s0)) / h
            rks.append(rk)
            d1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        hh = -h
        h_phi_1, b_h = torch.expm1(hh), torch.expm1(hh)
        x_t_ = st / s0 * x - at * h_phi_1 * m0
        pred = 0.5 * d1s[0] if d1s else 0           # order 2: rhos_p = [0.5]
        return x_t_ - at * b_h * pred

    def _correct(self, model_t, last_sample, order):
        m0, x = self.model_outputs[-1], last_sample
        st, s0 = self.sigmas[self.step_index], self.sigmas[self.step_index - 1]
        at, _ = self._als(st)
        h = self._lam(st) - self._lam(s0)
        rks, d1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lam(self.sigmas[self.step_index - (i + 1)]) - self._lam(s0)) / h
            rks.append(rk)
            d1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack(rks)
        hh = -h
        h_phi_1, b_h, R, b = self._rb(rks, hh, order)
        rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        x_t_ = st / s0 * x - at * h_phi_1 * m0
        corr = sum(rhos[k] * d1s[k] for k in range(len(d1s))) if d1s else 0
        return x_t_ - at * b_h * (corr + rhos[-1] * (model_t - m0))

    def step(self, model_output, sample):
        use_corr = self.step_index > 0 and self.last_sample is not None
        m_t = sample - self.sigmas[self.step_index] * model_output        # flow prediction -> x0 (sigma*v in the v dtype)
        if use_corr:
            sample = self._correct(m_t, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [m_t]
        this_order = min(self.order_max, self.n - self.step_index)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.order_max:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev


class CogDPMOracle:
    """Tensor-op by tensor-op restatement of diffusers' CogVideoXDPMScheduler (scaled-linear betas, SNR shift, zero
    terminal SNR, trailing spacing, v-prediction): get_variables / get_mult / step as published.  Third-party, from the
    published algorithm: UNPINNED.  Call site: pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:915-926."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.0120, snr_shift_scale=1.0):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        s = ac.sqrt()
        a0, at = s[0].clone(), s[-1].clone()
        self.alphas_cumprod = ((s - at) * (a0 / (a0 - at))) ** 2                 # rescale_zero_terminal_snr
        self.final_alpha_cumprod = torch.tensor(1.0, dtype=torch.float64)
        self.n_train = num_train_timesteps

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = torch.from_numpy(
            np.round(np.arange(self.n_train, 0, -self.n_train / n)).astype(np.int64) - 1)

    @staticmethod
    def _variables(a_t, a_p, a_b=None):
        lamb = ((a_t / (1 - a_t)) ** 0.5).log()
        lamb_next = ((a_p / (1 - a_p)) ** 0.5).log()
        h = lamb_next - lamb
        if a_b is not None:
            lamb_previous = ((a_b / (1 - a_b)) ** 0.5).log()
            return h, (lamb - lamb_previous) / h
        return h, None

    @staticmethod
    def _mult(h, r, a_t, a_p, a_b):
        mult1 = ((1 - a_p) / (1 - a_t)) ** 0.5 * (-h).exp()
        mult2 = (-2 * h).expm1() * a_p ** 0.5
        if a_b is not None:
            return mult1, mult2, 1 + 1 / (2 * r), 1 / (2 * r)
        return mult1, mult2

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, generator=None):
        prev_timestep = timestep - self.n_train // self.n
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        a_b = self.alphas_cumprod[timestep_back] if timestep_back is not None else None
        pred_original_sample = (a_t ** 0.5) * sample - ((1 - a_t) ** 0.5) * model_output          # v-prediction
        h, r = self._variables(a_t, a_p, a_b)
        mult = self._mult(h, r, a_t, a_p, a_b)
        mult_noise = (1 - a_p) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
        noise = torch.randn(sample.shape, generator=generator, dtype=sample.dtype)
        prev_sample = mult[0] * sample - mult[1] * pred_original_sample + mult_noise * noise
        if old_pred_original_sample is None or prev_timestep < 0:
            return prev_sample, pred_original_sample
        denoised_d = mult[2] * pred_original_sample - mult[3] * old_pred_original_sample
        noise = torch.randn(sample.shape, generator=generator, dtype=sample.dtype)
        x_advanced = mult[0] * sample - mult[1] * denoised_d + mult_noise * noise
        return x_advanced, pred_original_sample
