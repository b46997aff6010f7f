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


class CogVideoXDPMScheduler(CogVideoXDDIMScheduler):
    """diffusers' CogVideoXDPMScheduler (third-party, restated from the published SDE-DPM-Solver++(2M) form it
    implements -- parity unpinned): the class zai-org/CogVideoX-5b-I2V ships and the reference's validation
    instantiates (train_code/train_cogvideox_motion_FrameINO.py:692); the pipeline dispatches on it at
    pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:915-926 (extra arguments: the previous x0 prediction and the
    previous timestep).  Same alpha-bar table and trailing spacing as the DDIM class.

    Every step draws fresh noise: once on a first-order step (the first step, and the last where prev_timestep < 0),
    TWICE on a second-order step (diffusers computes the first-order `prev_sample` before it knows it will discard
    it) -- `noise(shape, ...)` below reproduces that stream for a given generator.  The update is linear, so
    `set_timesteps` folds each step's scalars (float64, as diffusers' 0-dim tensors) into a row of `coefs`:
    {sa, sb, m1, m2, m3, m4, mn, use_old} and the kernel is fino_cfg_dpm_step."""
    kind = "dpm"

    def set_timesteps(self, num_inference_steps, device=None, **unused):
        super().set_timesteps(num_inference_steps, device)
        n = self.config.num_train_timesteps
        ts = self.timesteps.tolist()
        rows = []
        for i, t in enumerate(ts):
            prev = t - n // num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
            a_b = self.alphas_cumprod[ts[i - 1]] if i > 0 else None
            lamb = ((a_t / (1 - a_t)) ** 0.5).log()
            lamb_next = ((a_p / (1 - a_p)) ** 0.5).log()
            h = lamb_next - lamb
            m1 = ((1 - a_p) / (1 - a_t)) ** 0.5 * (-h).exp()
            m2 = (-2 * h).expm1() * a_p ** 0.5
            mn = (1 - a_p) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5
            use_old = a_b is not None and prev >= 0
            m3 = m4 = 0.0
            if use_old:
                r = (lamb - ((a_b / (1 - a_b)) ** 0.5).log()) / h
                m3, m4 = float(1 + 1 / (2 * r)), float(1 / (2 * r))
            rows.append([float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(m1), float(m2), m3, m4, float(mn),
                         1.0 if use_old else 0.0])
        self.coefs = torch.tensor(rows, dtype=torch.float32, device=device)
        self._use_old = [r[7] != 0.0 for r in rows]          # host copy: the loop must not sync on the device table

    def draws(self, i):
        """number of standard-normal tensors diffusers' step consumes at step i"""
        return 2 if self._use_old[i] else 1

    def noise(self, i, shape, generator, device, dtype):
        """the draw step i uses (the LAST of its `draws(i)` draws), sampled like diffusers' randn_tensor: on the
        generator's device, then moved"""
        gdev = generator.device if generator is not None else device
        out = None
        for _ in range(self.draws(i)):
            out = torch.randn(shape, generator=generator, device=gdev, dtype=dtype)
        return out.to(device)


class UniPCMultistepScheduler:
    """diffusers' UniPCMultistepScheduler as Wan-AI/Wan2.2-TI2V-5B-Diffusers configures it (third-party, restated from
    the published algorithm -- parity unpinned): prediction_type="flow_prediction", use_flow_sigmas, flow_shift,
    solver_order=2, solver_type="bh2", predict_x0, lower_order_final, final_sigmas_type="zero".

    Every tensor operation of its step is linear in {sample, last_sample, model_outputs[-1], model_outputs[-2], v}, so
    `set_timesteps` folds each step's scalars into a coefficient row (fp64 on the host, one device table) and the
    update runs as ONE kernel, fused with CFG (fino_cfg_unipc_step)."""
    order = 1
    kind = "unipc"

    def __init__(self, num_train_timesteps=1000, solver_order=2, prediction_type="flow_prediction", flow_shift=5.0,
                 use_flow_sigmas=True, solver_type="bh2", predict_x0=True, lower_order_final=True,
                 final_sigmas_type="zero", **unused):
        if not (use_flow_sigmas and prediction_type == "flow_prediction" and predict_x0 and solver_type == "bh2"
                and solver_order in (1, 2) and final_sigmas_type == "zero"):
            raise NotImplementedError("only the Wan2.2 configuration of UniPC is implemented")
        self.config = _Cfg(num_train_timesteps=num_train_timesteps, solver_order=solver_order, flow_shift=flow_shift,
                           prediction_type=prediction_type, solver_type=solver_type)

    def set_timesteps(self, num_inference_steps, device=None, **unused):
        n_train, shift, order_max = self.config.num_train_timesteps, self.config.flow_shift, self.config.solver_order
        alphas = np.linspace(1, 1 / n_train, num_inference_steps + 1)
        sig = 1.0 - alphas
        sig = np.flip(shift * sig / (1 + (shift - 1) * sig))[:-1].copy()
        self.timesteps = torch.from_numpy((sig * n_train).copy()).to(device=device, dtype=torch.int64)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32).astype(np.float64)      # diffusers stores fp32 sigmas
        self.sigmas = torch.from_numpy(sig.astype(np.float32)).to(device)
        self.num_inference_steps = n = num_inference_steps

        lam = lambda s: np.log(1.0 - s) - np.log(s) if s > 0 else np.inf               # noqa: E731

        def bh(h, order, rk):
            """b-vector / R-matrix pieces of the bh2 update for step size h (predict_x0: hh = -h)."""
            hh = -h
            h_phi_1 = np.expm1(hh)
            b_h = np.expm1(hh)
            h_phi_k = h_phi_1 / hh - 1.0 if np.isfinite(hh) else -1.0
            b, fact = [], 1.0
            for i in range(1, order + 1):
                b.append(h_phi_k * fact / b_h)
                fact *= i + 1
                h_phi_k = h_phi_k / hh - 1.0 / fact if np.isfinite(hh) else -1.0 / fact
            return h_phi_1, b_h, np.array(b)

        rows, lower_order_nums, prev_order = [], 0, 0
        for i in range(n):
            s_i, s_n = sig[i], sig[i + 1]
            # ---- corrector (uses the order chosen at the previous step) ----
            use_corr, cx, c0, c1, ct = 0.0, 0.0, 0.0, 0.0, 0.0
            if i > 0:
                use_corr = 1.0
                s_p = sig[i - 1]
                h = lam(s_i) - lam(s_p)
                a_t = 1.0 - s_i
                if prev_order == 1:
                    h_phi_1, b_h, _ = bh(h, 1, None)
                    rho_t = 0.5
                    cx, c0, c1, ct = s_i / s_p, -a_t * h_phi_1 + a_t * b_h * rho_t, 0.0, -a_t * b_h * rho_t
                else:
                    rk = (lam(sig[i - 2]) - lam(s_p)) / h
                    h_phi_1, b_h, b = bh(h, 2, rk)
                    rho = np.linalg.solve(np.array([[1.0, 1.0], [rk, 1.0]]), b)
                    # x_t = s_i/s_p x - a_t h_phi_1 m0 - a_t B_h (rho0 (m1 - m0)/rk + rho1 (m_t - m0))
                    cx = s_i / s_p
                    c0 = -a_t * h_phi_1 + a_t * b_h * (rho[0] / rk + rho[1])
                    c1 = -a_t * b_h * rho[0] / rk
                    ct = -a_t * b_h * rho[1]
            # ---- predictor ----
            this_order = min(order_max, n - i)                        # lower_order_final
            this_order = min(this_order, lower_order_nums + 1)
            a_n = 1.0 - s_n
            if s_n > 0:
                h = lam(s_n) - lam(s_i)
                h_phi_1, b_h = np.expm1(-h), np.expm1(-h)
            else:
                h_phi_1, b_h = -1.0, -1.0                             # h = +inf at the final (sigma = 0) step
            px, p0, p1 = s_n / s_i, -a_n * h_phi_1, 0.0
            if this_order == 2:
                rk = (lam(sig[i - 1]) - lam(s_i)) / h
                # x_t = x_t_ - a_n B_h * 0.5 * (m_{i-1} - m_i)/rk
                p0 += a_n * b_h * 0.5 / rk
                p1 = -a_n * b_h * 0.5 / rk
            if lower_order_nums < order_max:
                lower_order_nums += 1
            prev_order = this_order
            rows.append([0.0, s_i, use_corr, cx, c0, c1, ct, px, p0, p1])
        self.coefs = torch.tensor(rows, dtype=torch.float32, device=device)       # column 0 (guidance) set by the caller
