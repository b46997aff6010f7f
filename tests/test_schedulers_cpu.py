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
