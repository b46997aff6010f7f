"""Host logic of the samplers: the coefficient-folded UniPC (what the fused kernel executes) equals the tensor-op
restatement of the published algorithm, and reduces to sane limits."""
import torch

from oracle.schedulers import UniPCOracle


def test_unipc_coefficient_form_equals_direct_algorithm():
    from frameino_amd.schedulers import UniPCMultistepScheduler
    n = 12
    sch = UniPCMultistepScheduler(flow_shift=5.0)
    sch.set_timesteps(n, device="cpu")
    orc = UniPCOracle(flow_shift=5.0)
    orc.set_timesteps(n)
    assert torch.equal(sch.timesteps, orc.timesteps)
    torch.testing.assert_close(sch.sigmas, orc.sigmas)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, dtype=torch.float64, generator=g).float()
    xo = x.clone()
    last = torch.zeros_like(x)
    m0 = torch.zeros_like(x)
    m1 = torch.zeros_like(x)
    for i in range(n):
        v = torch.randn(3, 5, generator=g)
        _, sigma, use_corr, cx, c0, c1, ct, px, p0, p1 = sch.coefs[i].tolist()
        mt = x - sigma * v
        xc = cx * last + c0 * m0 + c1 * m1 + ct * mt if use_corr else x
        x, last, m1, m0 = px * xc + p0 * mt + p1 * m0, xc, m0, mt
        xo = orc.step(v, xo)
        torch.testing.assert_close(x, xo, atol=2e-5, rtol=2e-5)
    # final sigma is 0: the last update returns the x0 prediction exactly
    torch.testing.assert_close(x, m0, atol=1e-6, rtol=1e-6)


def test_euler_and_ddim_tables():
    from frameino_amd.schedulers import CogVideoXDDIMScheduler, FlowMatchEulerDiscreteScheduler
    e = FlowMatchEulerDiscreteScheduler(shift=5.0)
    e.set_timesteps(50, device="cpu")
    assert e.timesteps.shape == (50,) and e.sigmas[-1] == 0 and torch.all(e.dts < 0)
    assert abs(e.sigmas[0].item() - 1.0) < 1e-6
    d = CogVideoXDDIMScheduler()
    d.set_timesteps(50, device="cpu")
    assert d.timesteps[0].item() == 999 and d.coefs.shape == (50, 4)
    assert abs(float(d.alphas_cumprod[-1])) < 1e-12             # zero terminal SNR


def test_dpm_coefficient_form_equals_direct_algorithm():
    """CogVideoXDPMScheduler: the folded rows fino_cfg_dpm_step executes vs the op-by-op restatement of the published
    step (oracle/schedulers.py::CogDPMOracle), including the RNG stream (two draws on second-order steps)."""
    from frameino_amd.schedulers import CogVideoXDPMScheduler
    from oracle.schedulers import CogDPMOracle
    n = 9
    sch = CogVideoXDPMScheduler()
    sch.set_timesteps(n, device="cpu")
    orc = CogDPMOracle()
    orc.set_timesteps(n)
    assert torch.equal(sch.timesteps, orc.timesteps) and sch.coefs.shape == (n, 8)
    assert [sch.draws(i) for i in range(n)] == [1] + [2] * (n - 2) + [1]
    g0 = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 4, generator=g0)
    xo, old_o = x.clone(), None
    x0_old = torch.zeros_like(x)
    ga, gb = torch.Generator().manual_seed(11), torch.Generator().manual_seed(11)
    ts = sch.timesteps.tolist()
    for i, t in enumerate(ts):
        v = torch.randn(2, 3, 4, generator=g0)
        sa, sb, m1, m2, m3, m4, mn, use_old = sch.coefs[i].tolist()
        x0 = sa * x - sb * v
        d = m3 * x0 - m4 * x0_old if use_old else x0
        x = m1 * x - m2 * d + mn * sch.noise(i, x.shape, ga, "cpu", x.dtype)
        x0_old = x0
        xo, old_o = orc.step(v, old_o, t, ts[i - 1] if i > 0 else None, xo, generator=gb)
        xo = xo.to(x.dtype)
        torch.testing.assert_close(x, xo, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(x0_old, old_o.to(x.dtype), atol=2e-5, rtol=2e-5)
    # the last step lands on alpha_bar = 1: no noise left, x equals the first-order x0 prediction
    assert abs(sch.coefs[-1, 6].item()) < 1e-6
    torch.testing.assert_close(x, x0_old, atol=1e-5, rtol=1e-5)


def test_samplers_converge_to_the_analytic_flow_of_a_gaussian_at_their_order():
    """A pin that owes nothing to diffusers: for Gaussian data N(mu, s^2) the posterior means are closed-form, so the
    probability-flow ODE has the exact solution x(sigma) = (1 - sigma) mu + sqrt((1 - sigma)^2 s^2 + sigma^2) z (flow matching;
    x(a) = sqrt(a) mu + sqrt(a s^2 + 1 - a) z on the variance-preserving path), which ends at mu + s z.  Driven by the exact
    velocity / v-prediction, the restated samplers must arrive there with the error of their order: flow-match Euler and DDIM
    halve it when the steps double, UniPC (order-2 predictor + corrector, as Wan2.2 configures it) cuts it ~ 8-fold."""
    import numpy as np
    from frameino_amd.schedulers import CogVideoXDDIMScheduler, FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
    mu, s = 0.7, 0.5
    z = torch.linspace(-2, 2, 9, dtype=torch.float64)
    end = mu + s * z

    def flow_v(x, sig):                                  # exact E[eps - x0 | x_sigma]
        a = 1 - sig
        x0 = mu + a * s * s * (x - a * mu) / (a * a * s * s + sig * sig)
        return (x - a * x0) / sig - x0

    def flow_x(sig):
        return (1 - sig) * mu + np.sqrt((1 - sig) ** 2 * s * s + sig * sig) * z

    def euler(n):
        e = FlowMatchEulerDiscreteScheduler(shift=5.0)
        e.set_timesteps(n, device="cpu")
        sg = e.sigmas.double()
        x = flow_x(sg[0].item())
        for i in range(n):
            x = x + (sg[i + 1] - sg[i]) * flow_v(x, sg[i].item())
        return (x - end).abs().max().item()

    def unipc(n):
        u = UniPCMultistepScheduler(flow_shift=5.0)
        u.set_timesteps(n, device="cpu")
        x = flow_x(u.sigmas[0].item())
        last, m0, m1 = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
        for i in range(n):                               # the coefficient rows the fused kernel executes
            _, sigma, use_corr, cx, c0, c1, ct, px, p0, p1 = u.coefs[i].double().tolist()
            mt = x - sigma * flow_v(x, sigma)
            xc = cx * last + c0 * m0 + c1 * m1 + ct * mt if use_corr else x
            x, last, m1, m0 = px * xc + p0 * mt + p1 * m0, xc, m0, mt
        return (x - end).abs().max().item()

    def ddim(n):
        d = CogVideoXDDIMScheduler(snr_shift_scale=1.0)
        d.set_timesteps(n, device="cpu")
        a0 = d.alphas_cumprod[d.timesteps[0]].item()
        x = np.sqrt(a0) * mu + np.sqrt(a0 * s * s + 1 - a0) * z
        for i, t in enumerate(d.timesteps.tolist()):
            a = d.alphas_cumprod[t].item()
            x0h = mu + np.sqrt(a) * s * s * (x - np.sqrt(a) * mu) / (a * s * s + 1 - a)
            v = np.sqrt(a) * (x - np.sqrt(a) * x0h) / np.sqrt(1 - a) - np.sqrt(1 - a) * x0h      # exact v-prediction
            sa, sb, ca, cb = d.coefs[i].double().tolist()
            x = ca * x + cb * (sa * x - sb * v)
        return (x - end).abs().max().item()

    e50, e100, e200 = euler(50), euler(100), euler(200)
    assert 1.7 < e50 / e100 < 2.3 and 1.7 < e100 / e200 < 2.3, (e50, e100, e200)
    d50, d100, d200 = ddim(50), ddim(100), ddim(200)
    assert 1.6 < d50 / d100 < 2.3 and 1.6 < d100 / d200 < 2.3, (d50, d100, d200)
    u16, u32, u64 = unipc(16), unipc(32), unipc(64)
    assert u16 / u32 > 4.0 and u32 / u64 > 7.0 and u64 < e200 / 5, (u16, u32, u64, e200)


def test_dpm_sde_sampler_draws_the_gaussian_it_is_given_the_exact_score_of():
    """The stochastic sampler cannot be checked path by path, but in law: driven by the exact v-prediction of Gaussian data
    N(mu, s^2) on the variance-preserving path, the folded SDE-DPM-Solver++(2M) rows must produce samples whose mean is mu
    and whose standard deviation approaches s as the steps grow (a wrong noise scale, history weight or exponential factor
    shifts one or the other)."""
    import numpy as np
    from frameino_amd.schedulers import CogVideoXDPMScheduler
    mu, s, m = 0.7, 0.5, 200000

    def run(n):
        d = CogVideoXDPMScheduler(snr_shift_scale=1.0)
        d.set_timesteps(n, device="cpu")
        g = torch.Generator().manual_seed(5)
        a0 = d.alphas_cumprod[d.timesteps[0]].item()
        x = np.sqrt(a0) * mu + np.sqrt(a0 * s * s + 1 - a0) * torch.randn(m, generator=g, dtype=torch.float64)
        x0_old = torch.zeros_like(x)
        for i, t in enumerate(d.timesteps.tolist()):
            a = d.alphas_cumprod[t].item()
            x0h = mu + np.sqrt(a) * s * s * (x - np.sqrt(a) * mu) / (a * s * s + 1 - a)
            v = np.sqrt(a) * (x - np.sqrt(a) * x0h) / np.sqrt(1 - a) - np.sqrt(1 - a) * x0h
            sa, sb, m1, m2, m3, m4, mn, use_old = d.coefs[i].double().tolist()
            x0 = sa * x - sb * v
            x = m1 * x - m2 * (m3 * x0 - m4 * x0_old if use_old else x0) + mn * torch.randn(m, generator=g, dtype=torch.float64)
            x0_old = x0
        return x.mean().item(), x.std().item()

    (m10, s10), (m25, s25), (m50, s50) = run(10), run(25), run(50)
    assert abs(m10 - mu) < 5e-3 and abs(m25 - mu) < 5e-3 and abs(m50 - mu) < 5e-3, (m10, m25, m50)
    assert abs(s10 - s) > abs(s25 - s) > abs(s50 - s) and abs(s50 - s) < 0.02, (s10, s25, s50)
