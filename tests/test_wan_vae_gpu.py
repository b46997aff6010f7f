"""Wan VAE on the HIP path: conv kernel vs F.conv3d for every geometry the VAE uses, and whole-model encode / decode
vs the golden vectors recorded from the reference's chunked streaming run (bf16 storage vs the fp32 reference:
stated tolerance rel-RMS <= 3e-2 on the latent moments, PSNR >= 35 dB on the decoded video in [-1, 1])."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.parity import rel_rms

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pack(w, cin_pad, cout_pad):
    co, ci, kt, kh, kw = w.shape
    w2 = torch.zeros(cout_pad, kt * kh * kw, cin_pad)
    w2[:co, :, :ci] = w.permute(0, 2, 3, 4, 1).reshape(co, kt * kh * kw, ci)
    return w2.reshape(cout_pad, -1)


@pytest.mark.parametrize("name,ci,co,k,stride,pad,up,thw", [
    ("causal3x3x3", 48, 72, (3, 3, 3), (1, 1, 1), (2, 1, 1), False, (5, 6, 7)),
    ("1x1x1", 100, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (3, 5, 4)),
    ("time_conv", 64, 128, (3, 1, 1), (1, 1, 1), (2, 0, 0), False, (4, 5, 6)),
    ("up2x+conv2d", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), True, (3, 5, 6)),
    ("down conv2d s2", 40, 40, (1, 3, 3), (1, 2, 2), (0, 0, 0), False, (3, 8, 10)),
    ("down time s2", 64, 64, (3, 1, 1), (2, 1, 1), (0, 0, 0), False, (9, 4, 5)),
])
def test_conv3d_geometries_vs_torch(name, ci, co, k, stride, pad, up, thw):
    from frameino_amd import ops
    from frameino_amd.autoencoder_kl_wan import cpad
    g = torch.Generator().manual_seed(1)
    t, h, w = thw
    x = torch.randn(1, ci, t, h, w, generator=g)
    wt = torch.randn(co, ci, *k, generator=g) / math.sqrt(ci * k[0] * k[1] * k[2])
    b = torch.randn(co, generator=g) * 0.1
    xb, wb, bb = x.bfloat16().float(), wt.bfloat16().float(), b.bfloat16().float()
    xin = xb
    if up:
        xin = F.interpolate(xb[0].permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact")
        xin = xin.permute(1, 0, 2, 3)[None]
    if name == "down conv2d s2":
        xin = F.pad(xin, (0, 1, 0, 1))
        ref = F.conv3d(xin, wb, bb, stride=stride)
    else:
        ref = F.conv3d(F.pad(xin, (pad[2], pad[2], pad[1], pad[1], pad[0], 0)), wb, bb, stride=stride)
    xcl = torch.zeros(t, h, w, cpad(ci))
    xcl[..., :ci] = xb[0].permute(1, 2, 3, 0)
    w2 = _pack(wb, cpad(ci), cpad(co))
    b2 = torch.zeros(cpad(co))
    b2[:co] = bb
    out_thw = tuple(ref.shape[2:])
    y = ops.conv3d_cl(xcl.to(DEV).bfloat16(), w2.to(DEV).bfloat16(), b2.to(DEV).bfloat16(), k, stride, pad, out_thw,
                      up)
    got = y[..., :co].permute(3, 0, 1, 2).float().cpu()[None]
    assert got.shape == ref.shape
    assert rel_rms(got, ref) < 6e-3, (name, rel_rms(got, ref))
    if cpad(co) > co:
        assert y[..., co:].abs().max().item() == 0              # pad channels stay exactly zero
    # fused residual epilogue
    r = torch.randn_like(y)
    y2 = ops.conv3d_cl(xcl.to(DEV).bfloat16(), w2.to(DEV).bfloat16(), b2.to(DEV).bfloat16(), k, stride, pad, out_thw,
                       up, residual=r)
    assert rel_rms(y2, (y.float() + r.float())) < 4e-3


def _vae(golden, name, prefix=""):
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    cfg, sd, a = golden(name)
    if prefix:
        cfg = {k[4:]: v for k, v in cfg.items() if k.startswith("vae_")}
        sd = {k[4:]: v for k, v in sd.items() if k.startswith("vae.")}
    keys = ("base_dim", "decoder_base_dim", "z_dim", "dim_mult", "num_res_blocks", "temperal_downsample",
            "latents_mean", "latents_std", "is_residual", "in_channels", "out_channels", "patch_size",
            "scale_factor_temporal", "scale_factor_spatial")
    kw = {k: (list(cfg[k]) if isinstance(cfg[k], (list, tuple)) else cfg[k]) for k in keys}
    kw["is_residual"] = bool(kw["is_residual"])
    vae = AutoencoderKLWan(**kw).to(DEV)
    vae.load_reference_state_dict(sd, dtype=torch.bfloat16)
    return vae, a


def psnr(a, b, peak=2.0):
    mse = (a.float().cpu() - b.float().cpu()).pow(2).mean().item()
    return 10 * math.log10(peak * peak / max(mse, 1e-20))


def test_decode_matches_reference_streaming_decode(golden):
    vae, a = _vae(golden, "wan_vae_tiny")
    for nl in (1, 2, 3):
        out = vae.decode(a[f"dec_in_{nl}"].to(DEV), return_dict=False)[0]
        ref = a[f"dec_out_{nl}"]
        assert out.shape == ref.shape
        assert psnr(out, ref) > 35.0, (nl, psnr(out, ref))


def test_encode_matches_reference_streaming_encode(golden):
    vae, a = _vae(golden, "wan_vae_tiny")
    for nf in (1, 5, 9):
        post = vae.encode(a[f"enc_in_{nf}"].to(DEV)).latent_dist
        ref = a[f"enc_out_{nf}"]
        assert post.parameters.shape == ref.shape
        z = ref.shape[1] // 2
        assert rel_rms(post.mode(), ref[:, :z]) < 3e-2, (nf, rel_rms(post.mode(), ref[:, :z]))
        assert rel_rms(post.parameters, ref) < 3e-2


def test_pipeline_conditions_and_video_of_the_recorded_reference_run(golden):
    """prepare_latents (:400-553) + decode (:916-927) of the reference pipeline run, on the HIP VAE."""
    vae, a = _vae(golden, "wan_pipe_tiny", prefix="vae")
    z = vae.config.z_dim
    mean = torch.tensor(vae.config.latents_mean).view(1, z, 1, 1, 1).to(DEV)
    inv_std = (1.0 / torch.tensor(vae.config.latents_std)).view(1, z, 1, 1, 1).to(DEV)
    traj = a["traj"].unsqueeze(0).permute(0, 2, 1, 3, 4).to(DEV)
    tl = (vae.encode(traj).latent_dist.mode() - mean) * inv_std
    assert rel_rms(tl, a["traj_latents"][:, :, :tl.shape[2]]) < 3e-2
    video = vae.decode(a["out_latents"].to(DEV) / inv_std + mean, return_dict=False)[0]
    got = (video / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 4, 1)
    assert psnr(got, a["out_video"], peak=1.0) > 35.0


def test_full_width_vae_vs_oracle_small_video():
    """The real Wan2.2 VAE widths (base 160 / decoder 256, z = 48, 2x2 patchify, residual blocks) with seeded random
    weights on a small video (9 frames 64x96 -> 3 latent frames 4x6 and back), against the oracle in fp32: the wide
    channel counts (up to 1024) and every conv geometry at their real K, which the tiny golden cannot reach."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    from oracle import wan_vae as V
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=5, device=DEV)
    sd = {k: v.float().cpu() for k, v in vae._sd.items()}
    cfg = dict(WAN22_VAE_CFG)
    g = torch.Generator().manual_seed(6)
    vid = torch.rand(1, 3, 9, 64, 96, generator=g) * 2 - 1
    ref_z = V.wan_vae_encode(sd, cfg, vid)                       # moments [1, 96, 3, 4, 6]
    post = vae.encode(vid.to(DEV)).latent_dist
    assert post.parameters.shape == ref_z.shape
    r_enc = rel_rms(post.parameters, ref_z)
    z = torch.randn(1, 48, 3, 4, 6, generator=g)
    ref_v = V.wan_vae_decode(sd, cfg, z)
    out = vae.decode(z.to(DEV), return_dict=False)[0]
    assert out.shape == ref_v.shape == (1, 3, 9, 64, 96)
    p = psnr(out, ref_v)
    print(f"full-width VAE: encode rel-RMS {r_enc:.4f}, decode PSNR {p:.1f} dB")
    assert r_enc < 3e-2 and p > 35.0


@pytest.mark.parametrize("chunk", [1, 2, 3, 5])
def test_time_chunked_decoder_tail_is_bit_identical_to_whole_sequence(chunk):
    """`decode_chunk_frames`: the decoder's tail (the blocks after the last temporal upsampling + the head) run chunk
    by chunk in time with the last two input frames of every causal conv carried over (what the reference's feat_cache
    streaming does, :350-358) -- a memory schedule only: every output element sees the same taps in the same order."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    cfg = dict(base_dim=16, decoder_base_dim=16, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
               temperal_downsample=[False, True, True], is_residual=True, in_channels=12, out_channels=12, patch_size=2,
               scale_factor_temporal=4, scale_factor_spatial=16)
    vae = AutoencoderKLWan(**cfg).random_init_(seed=4, device=DEV)
    z = torch.randn(1, 4, 4, 3, 5, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))    # -> 13 frames
    vae.decode_chunk_frames = 0
    whole = vae.decode(z, return_dict=False)[0]
    vae.decode_chunk_frames = chunk
    chunked = vae.decode(z, return_dict=False)[0]
    assert whole.shape == chunked.shape == (1, 3, 13, 48, 80)
    assert torch.equal(whole, chunked)


def test_full_size_decode_vs_oracle_fp32_on_device():
    """the reference app runs this VAE in fp32 (app.py:157), the mirror's convolutions compute in bf16 -- the
    number at SIZE: 5 latent frames at 704x1280 (latent 44x80 -> 17 frames of 704x1280), the real Wan2.2 VAE widths, seeded
    random weights, against oracle/wan_vae.py executed in fp32 on the device."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    from oracle import wan_vae as V
    from tests.parity import record
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device=DEV)
    sd = {k: v.float() for k, v in vae._sd.items()}
    z = torch.randn(1, 48, 5, 44, 80, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    with torch.no_grad():
        out = vae.decode(z, return_dict=False)[0]
        ref = V.wan_vae_decode(sd, dict(WAN22_VAE_CFG), z)
    torch.cuda.synchronize()
    assert out.shape == ref.shape == (1, 3, 17, 704, 1280) and torch.isfinite(out).all()
    p = psnr(out, ref)
    r = rel_rms(out, ref)
    inside = ref.abs() < 0.98                                       # where the [-1, 1] clamp (:1221) is not active
    r_in = rel_rms(out[inside], ref[inside])
    record("wan_vae_decode_full_size_5_latent_frames_704x1280", "PSNR dB hip bf16 vs oracle fp32 on device (higher is better)",
           p, 38.0, lower_is_better=False)
    record("wan_vae_decode_full_size_5_latent_frames_704x1280", "rel_rms (all pixels / unclamped pixels: "
           f"{r_in:.4f})", r, 2.5e-2)
    assert p > 38.0 and r < 2.5e-2, (p, r)


# ------------------------------------------------------------------------------------------------ fp32-compute mode (round 5)
@pytest.mark.parametrize("planes,bound", [(3, 2e-6), (2, 6e-5)])
@pytest.mark.parametrize("name,ci,co,k,stride,pad,up,thw", [
    ("causal3x3x3", 48, 72, (3, 3, 3), (1, 1, 1), (2, 1, 1), False, (5, 6, 7)),
    ("1x1x1", 100, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, (3, 5, 4)),
    ("up2x+conv2d", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), True, (3, 5, 6)),
    ("down conv2d s2", 40, 40, (1, 3, 3), (1, 2, 2), (0, 0, 0), False, (3, 8, 10)),
    ("down time s2", 64, 64, (3, 1, 1), (2, 1, 1), (0, 0, 0), False, (9, 4, 5)),
    ("wide K", 512, 256, (3, 3, 3), (1, 1, 1), (2, 1, 1), False, (3, 9, 10)),
])
def test_split_bf16_conv_vs_torch_fp64(name, ci, co, k, stride, pad, up, thw, planes, bound):
    """fino_conv3d_split: fp32 activations and weights as bf16 planes, the cross terms >= 2^-16 relative accumulated on the matrix
    pipe -- against the same convolution in fp64 on UNROUNDED fp32 operands.  Three planes: the error of an fp32 convolution
    (torch's own fp32 conv measures the same); two planes: ~2^-16."""
    from frameino_amd import ops
    from frameino_amd.autoencoder_kl_wan import cpad
    g = torch.Generator().manual_seed(1)
    t, h, w = thw
    x = torch.randn(1, ci, t, h, w, generator=g)
    wt = torch.randn(co, ci, *k, generator=g) / math.sqrt(ci * k[0] * k[1] * k[2])
    b = torch.randn(co, generator=g) * 0.1
    xin = x.double()
    if up:
        xin = F.interpolate(xin[0].permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest-exact").permute(1, 0, 2, 3)[None]
    if name == "down conv2d s2":
        ref = F.conv3d(F.pad(xin, (0, 1, 0, 1)), wt.double(), b.double(), stride=stride)
    else:
        ref = F.conv3d(F.pad(xin, (pad[2], pad[2], pad[1], pad[1], pad[0], 0)), wt.double(), b.double(), stride=stride)
    xcl = torch.zeros(t, h, w, cpad(ci))
    xcl[..., :ci] = x[0].permute(1, 2, 3, 0)
    w2 = _pack(wt, cpad(ci), cpad(co))                                       # [Cout_pad, taps * Cin_pad] fp32
    taps = k[0] * k[1] * k[2]
    w6 = ops.split_bf16(w2.reshape(cpad(co) * taps, cpad(ci)).to(DEV).contiguous(), "W", planes).reshape(cpad(co), -1)
    b2 = torch.zeros(cpad(co))
    b2[:co] = b
    xp = ops.split_bf16(xcl.to(DEV), "planes", planes)
    assert xp.shape == (t, h, w, planes * cpad(ci)) and xp.dtype == torch.bfloat16
    # the planes add up to the fp32 value (three planes: exactly, up to the last bit of a 24-bit significand)
    back = xp.float().view(t, h, w, planes, cpad(ci)).sum(3).cpu()
    assert (back - xcl).abs().max().item() <= (2.0 ** -22 if planes == 3 else 2.0 ** -15) * xcl.abs().max().item()
    out_thw = tuple(ref.shape[2:])
    y = ops.conv3d_split_cl(xp, w6, b2.to(DEV), k, planes, stride, pad, out_thw, up)
    assert y.dtype == torch.float32
    got = y[..., :co].permute(3, 0, 1, 2).double().cpu()[None]
    e = rel_rms(got, ref)
    e32 = rel_rms(F.conv3d(F.pad(xin.float(), (0, 1, 0, 1)) if name == "down conv2d s2" else
                           F.pad(xin.float(), (pad[2], pad[2], pad[1], pad[1], pad[0], 0)), wt, b, stride=stride).double(), ref)
    print(f"{name} planes={planes}: split-bf16 conv rel-RMS {e:.2e} vs fp64 (torch fp32 conv on the CPU: {e32:.2e})")
    assert got.shape == ref.shape and e < bound, (name, planes, e)
    if cpad(co) > co:
        assert y[..., co:].abs().max().item() == 0
    r = torch.randn(y.shape, device=DEV)
    y2 = ops.conv3d_split_cl(xp, w6, b2.to(DEV), k, planes, stride, pad, out_thw, up, residual=r)
    assert torch.equal(y2, y + r)                                            # one fp32 add in the epilogue


def test_split_bf16_plain_gemm_fp32_output():
    """fino_gemm with FINO_EPI_F32 / FINO_EPI_F32_RESIDUAL on operands expanded one plane per product (the mid-block attention's
    matrix products in the fp32-compute mode): ragged M, N = 8 and N = 3520-ish, against fp64."""
    from frameino_amd import ops
    g = torch.Generator().manual_seed(3)
    for m, n, kdim in ((300, 1024, 1024), (3520, 3520, 1024), (77, 8, 64)):
        a = torch.randn(m, kdim, generator=g).to(DEV)
        w = (torch.randn(n, kdim, generator=g) / math.sqrt(kdim)).to(DEV)
        b = torch.randn(n, generator=g).to(DEV)
        r = torch.randn(m, n, generator=g).to(DEV)
        ref = a.double() @ w.double().t() + b.double()
        for planes, bound in ((3, 2e-6), (2, 6e-5)):
            got = ops.gemm_f32(ops.split_bf16(a, "A", planes), ops.split_bf16(w, "W", planes), b)
            assert got.dtype == torch.float32 and rel_rms(got.double(), ref) < bound, (m, n, planes, rel_rms(got.double(), ref))
            got2 = ops.gemm_f32(ops.split_bf16(a, "A", planes), ops.split_bf16(w, "W", planes), b, residual=r)
            assert torch.equal(got2, got + r)


def test_fp32_compute_mode_tiny_vae_vs_reference_golden(golden):
    """`set_compute_dtype(torch.float32)` on the tiny VAE against the reference's own fp32 streaming run (the golden fixture):
    decode 1 / 2 / 3 latent frames, encode 1 / 5 / 9 frames -- at fp32 accuracy, where the bf16 mode is held to 35 dB / 3e-2."""
    from tests.parity import record
    vae, a = _vae(golden, "wan_vae_tiny")
    vae.set_compute_dtype(torch.float32)
    assert vae.compute_dtype == torch.float32
    worst_d = worst_e = worst_dr = 0.0
    for nl in (1, 2, 3):
        out = vae.decode(a[f"dec_in_{nl}"].to(DEV), return_dict=False)[0]
        ref = a[f"dec_out_{nl}"]
        assert out.shape == ref.shape and out.dtype == torch.float32
        worst_d = max(worst_d, (out.cpu() - ref).abs().max().item())
        worst_dr = max(worst_dr, rel_rms(out, ref))
    for nf in (1, 5, 9):
        post = vae.encode(a[f"enc_in_{nf}"].to(DEV)).latent_dist
        ref = a[f"enc_out_{nf}"]
        assert post.parameters.shape == ref.shape
        worst_e = max(worst_e, rel_rms(post.parameters, ref))
    # (2e-5 max-abs is the bar the fp32 ORACLE is held to against the same fixture, tests/test_oracle_golden.py: two fp32
    # evaluations of one network in different summation orders)
    record("wan_vae_tiny_fp32_compute[decode]", f"max-abs vs the reference's fp32 streaming decode (video in [-1, 1]; rel_rms "
           f"{worst_dr:.2e})", worst_d, 2e-5)
    record("wan_vae_tiny_fp32_compute[decode rel_rms]", "rel_rms vs the reference's fp32 streaming decode", worst_dr, 1e-5)
    record("wan_vae_tiny_fp32_compute[encode]", "rel_rms of the moments vs the reference's fp32 streaming encode", worst_e, 1e-5)
    assert worst_d < 2e-5 and worst_dr < 1e-5 and worst_e < 1e-5, (worst_d, worst_dr, worst_e)
    # ... and the time-chunked tail stays bit-identical in this mode too
    vae.decode_chunk_frames = 0
    z = a["dec_in_3"].to(DEV)
    whole = vae.decode(z, return_dict=False)[0]
    vae.decode_chunk_frames = 2
    assert torch.equal(whole, vae.decode(z, return_dict=False)[0])
    # ... and so do the horizontal strips a convolution over the kernel's 2 GiB gather span is cut into (one layer of the
    # 704 x 1280 decode): forced here by a tiny limit
    vae.decode_chunk_frames = 0
    vae._span_limit = 1 << 12
    assert torch.equal(whole, vae.decode(z, return_dict=False)[0])
    vae._span_limit = 1 << 31
    vae.set_compute_dtype(torch.bfloat16)
    assert vae.compute_dtype == torch.bfloat16 and psnr(vae.decode(z, return_dict=False)[0], a["dec_out_3"]) > 35.0


@pytest.mark.parametrize("planes,min_psnr,max_rel", [(3, 80.0, 1e-4), (2, 70.0, 4e-4)])
def test_full_size_decode_fp32_compute_vs_oracle_fp32_on_device(planes, min_psnr, max_rel):
    """the reference app decodes in fp32 (app.py:157).  5 latent frames at 704x1280, the real Wan2.2 VAE widths,
    seeded random weights: `set_compute_dtype(torch.float32)` against oracle/wan_vae.py executed in fp32 on the device (whose own
    convolutions are an fp32 library's: two fp32 computations of the same network, so the bound is what THEY differ by)."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    from oracle import wan_vae as V
    from tests.parity import record
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device=DEV)
    vae.set_compute_dtype(torch.float32, planes=planes)
    sd = {k: v.float() for k, v in vae._sd.items()}
    z = torch.randn(1, 48, 5, 44, 80, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    import time
    with torch.no_grad():
        out = vae.decode(z, return_dict=False)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = vae.decode(z, return_dict=False)[0]
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        ref = V.wan_vae_decode(sd, dict(WAN22_VAE_CFG), z)
    torch.cuda.synchronize()
    assert out.shape == ref.shape == (1, 3, 17, 704, 1280) and torch.isfinite(out).all()
    p, r = psnr(out, ref), rel_rms(out, ref)
    print(f"fp32-compute decode, {planes} planes: PSNR {p:.1f} dB, rel-RMS {r:.2e}, {dt_s:.2f} s for 17 frames")
    record(f"wan_vae_decode_fp32_compute_{planes}planes_5_latent_frames_704x1280", "PSNR dB vs oracle fp32 on device (higher is better)",
           p, min_psnr, lower_is_better=False)
    record(f"wan_vae_decode_fp32_compute_{planes}planes_5_latent_frames_704x1280", f"rel_rms ({dt_s:.2f} s for 17 frames)", r, max_rel)
    assert p > min_psnr and r < max_rel, (p, r)


@pytest.mark.parametrize("planes,max_rel", [(0, 2.5e-2), (3, 1e-4)])
def test_full_size_encode_vs_oracle_fp32_on_device(planes, max_rel):
    """The encoder at SIZE (app.py:586-590 encodes the 704x1280 conditions with this VAE in fp32): 9 frames of 704x1280 -> 3 latent
    frames, the real Wan2.2 VAE widths, seeded random weights, bf16 compute and `set_compute_dtype(torch.float32)`, against
    oracle/wan_vae.py executed in fp32 on the device; the compared quantity is the posterior's moments (mean | logvar)."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    from oracle import wan_vae as V
    from tests.parity import record
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device=DEV)
    if planes:
        vae.set_compute_dtype(torch.float32, planes=planes)
    sd = {k: v.float() for k, v in vae._sd.items()}
    x = torch.rand(1, 3, 9, 704, 1280, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)) * 2 - 1
    with torch.no_grad():
        got = vae.encode(x).latent_dist.parameters
        ref = V.wan_vae_encode(sd, dict(WAN22_VAE_CFG), x)
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (1, 96, 3, 44, 80) and torch.isfinite(got).all()
    r = rel_rms(got, ref)
    name = "bf16" if not planes else f"fp32_compute_{planes}planes"
    print(f"encode 9 frames 704x1280, {name}: rel-RMS of the moments {r:.2e}")
    record(f"wan_vae_encode_{name}_9_frames_704x1280", "rel_rms of the moments vs oracle fp32 on device", r, max_rel)
    assert r < max_rel, r


# ------------------------------------------------------------------------------------------------ decoder tail in slabs (round 6)
@pytest.mark.parametrize("count,latent_frames,fp32", [(4, 13, False), (2, 5, False), (3, 5, False), (8, 3, False), (2, 3, True)],
                         ids=["4 slabs, 49 frames", "2 slabs", "3 ragged slabs", "8 slabs", "2 slabs fp32-compute"])
def test_decoder_tail_in_slabs_is_bit_identical_to_the_whole_decode_at_704x1280(count, latent_frames, fp32):
    """N ranks of a node decode N horizontal slabs of the frame (frameino_amd/parallel.py::sharded_vae_decode): the blocks up to the
    last temporal upsampling whole, the tail (up_blocks.2 / 3 + head: 74 % of the FLOPs, 15 convolutions, no attention) on the slab
    + 10 halo rows each side.  Simulated here as N calls on one GPU: the slabs put together must EQUAL `decode(z)`, bit for bit, at
    the bench's 704 x 1280 -- whole slabs, ragged ones (176 rows / 3), the fp32-compute mode."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device=DEV)
    if fp32:
        vae.set_compute_dtype(torch.float32)
    z = torch.randn(1, 48, latent_frames, 44, 80, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    whole = vae.decode(z, return_dict=False)[0]
    assert whole.shape == (1, 3, 1 + 4 * (latent_frames - 1), 704, 1280)
    assert vae._tail_halo_rows(2, 4) == (10, 4)
    rows_seen = 0
    for i in range(count):
        part, (r0, r1, per, height) = vae.decode_slab(z, i, count)
        assert height == 704 and r0 == rows_seen and per == -(-176 // count) * 4
        assert part.shape == (1, 3, whole.shape[2], r1 - r0, 1280)
        assert torch.equal(part, whole[:, :, :, r0:r1]), (i, float((part - whole[:, :, :, r0:r1]).abs().max()))
        rows_seen = r1
        del part
    assert rows_seen == 704
    # a slab count larger than anything sensible still tiles the frame (empty slabs are None)
    got = [vae.decode_slab(z[:, :, :1], i, 200)[0] for i in (0, 175, 176, 199)]
    assert got[0] is not None and got[1] is not None and got[2] is None and got[3] is None


def test_decoder_tail_in_eight_slabs_at_config4_size_1024x1792():
    """BASELINE config 4's frame (1024 x 1792: 256 rows of decoder-tail input, 32 per rank + 10 halo rows each side) on the 8 ranks it
    names, 3 latent frames: every slab bit-equal to its rows of the whole decode"""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=9, device=DEV)
    z = torch.randn(1, 48, 3, 64, 112, device=DEV, generator=torch.Generator(device=DEV).manual_seed(10))
    whole = vae.decode(z, return_dict=False)[0]
    assert whole.shape == (1, 3, 9, 1024, 1792)
    for i in range(8):
        part, (r0, r1, per, height) = vae.decode_slab(z, i, 8)
        assert (r0, r1, per, height) == (128 * i, 128 * (i + 1), 128, 1024)
        assert torch.equal(part, whole[:, :, :, r0:r1])
        del part


@pytest.mark.parametrize("count,frames,fp32", [(4, 49, False), (2, 9, False), (3, 9, False), (8, 5, False), (2, 5, True)],
                         ids=["4 slabs, 49 frames", "2 slabs", "3 ragged slabs", "8 slabs", "2 slabs fp32-compute"])
def test_encoder_head_in_slabs_is_bit_identical_to_the_whole_encode_at_704x1280(count, frames, fp32):
    """The trajectory-video encode on N ranks (parallel.sharded_vae_encode): conv_in + down_blocks.0 / .1 on horizontal slabs with a
    16-row halo (a multiple of 4, so that a slab keeps the frame's phase through both stride-2 convolutions), the slabs' activations put
    together, the rest of the encoder on the whole tensor.  Simulated as N calls on one GPU: the assembled activation must resume to
    EXACTLY the moments of `encode(x)`."""
    from frameino_amd.autoencoder_kl_wan import AutoencoderKLWan
    from frameino_amd.configs import WAN22_VAE_CFG
    vae = AutoencoderKLWan(**WAN22_VAE_CFG).random_init_(seed=7, device=DEV)
    if fp32:
        vae.set_compute_dtype(torch.float32)
    x = torch.rand(1, 3, frames, 704, 1280, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8)) * 2 - 1
    whole = vae.encode(x).latent_dist.parameters
    assert vae._enc_halo_rows(2) == (16, 16)
    parts, rows_seen = [], 0
    for i in range(count):
        part, (a, b, per, hk) = vae.encode_slab(x, i, count)
        assert hk == 88 and a == rows_seen and per == -(-88 // count) and part.shape[1] == b - a
        parts.append(part)
        rows_seen = b
    assert rows_seen == 88
    post = vae.encode_resume(torch.cat(parts, dim=1)).latent_dist
    assert torch.equal(post.parameters, whole)
