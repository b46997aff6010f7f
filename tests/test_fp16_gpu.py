"""fp16 as a first-class dtype (round 6).

fp16 is what the reference's canonical callers load -- `app.py:156` (`WanTransformer3DModel.from_pretrained(...,
torch_dtype=torch.float16)` + an fp32 VAE, `app.py:157`) and `test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-94,106`
(transformer, text encoder and VAE all fp16) -- and the dtype BASELINE.json's north star states its tolerance in.

Checked here, all on the HIP path in fp16 storage (fp32 accumulation, the reference's fp32 islands):
  * the tiny models / block / attention calls against the REFERENCE'S OWN fp16 runs (tests/golden/*_fp16*.npz, recorded by
    tools/golden/make_golden.py from /root/reference on the CPU) and against its fp32 runs;
  * SATURATION: with one weight scaled until a handful of intermediates leave fp16 range, the HIP block returns inf / nan /
    finite in the same elements as the reference's fp16 run (the fused epilogues round where the reference rounds);
  * both drop-in `__call__`s in the callers' precision mix (fp16 DiT + fp32 VAE for Wan; all-fp16 for CogVideoX);
  * the epilogues / sampler chains at kernel level against torch fp16 arithmetic on the device, overflow included;
  * the re-associated text out-projection P.(V W_o^T) in fp16, and its fall-back when V W_o^T leaves the range;
  * the CogVideoX VAE in fp16 against the oracle.
Stated tolerances: fp16 HIP vs the reference's fp32 run rel-RMS <= 5e-3 on the tiny DiTs (bf16: 3e-2), vs the reference's own
fp16 run <= 2.5e-3; full-size 2-layer forwards vs the fp32 oracle <= 2e-3 (tests/test_fullsize_oracle_gpu.py)."""
import math

import pytest
import torch

from tests.parity import hip_wan_model, record, rel_rms

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = torch.float16


def _classes_differ(out, ref):
    """number of elements that are nan / +inf / -inf / finite on one side and something else on the other"""
    out, ref = out.float().cpu(), ref.float().cpu()
    return int((~((torch.isnan(out) == torch.isnan(ref)) & ((out == math.inf) == (ref == math.inf))
                  & ((out == -math.inf) == (ref == -math.inf)))).sum())


# ------------------------------------------------------------------------------------------------ Wan DiT
def test_wan_fp16_model_vs_the_reference_fp16_and_fp32_runs(golden):
    cfg, sd, a = golden("wan_dit_tiny")
    _, _, h = golden("wan_dit_tiny_fp16")
    m = hip_wan_model(cfg, sd, DEV, dtype=H)
    assert m.dtype == H
    x, txt = a["x"].to(DEV).half(), a["txt"].to(DEV).half()
    for ts, y in (("ts_scalar", "y_scalar"), ("ts_tok", "y_tok"), ("ts_many", "y_many")):
        out = m(x, a[ts].to(DEV), txt, return_dict=False)[0]
        assert out.dtype == H and out.shape == a[y].shape
        r32, rh, ref_h = rel_rms(out, a[y]), rel_rms(out, h[y + "_fp16"]), rel_rms(h[y + "_fp16"], a[y])
        record(f"fp16_wan_dit_tiny[{ts}]", f"rel_rms hip fp16 vs reference fp32 run (reference's own fp16 run: {ref_h:.5f})", r32, 5e-3)
        record(f"fp16_wan_dit_tiny[{ts} vs ref fp16]", "rel_rms hip fp16 vs reference fp16 run", rh, 2.5e-3)
        assert r32 < 5e-3 and rh < 2.5e-3, (ts, r32, rh)
        assert r32 < 1.5 * ref_h + 5e-4, (r32, ref_h)           # no further from fp32 than the reference's own fp16 arithmetic


def test_wan_fp16_attention_plugin_calls_vs_the_reference_fp16_run(golden):
    """`Attention.forward` -> MI355WanAttnProcessor on fp16 tensors: the self- and the cross-attention call of wan_block_tiny"""
    cfg, sd, a = golden("wan_block_tiny")
    full_cfg, full_sd, _ = golden("wan_dit_tiny")
    _, _, h = golden("wan_dit_tiny_fp16")
    m = hip_wan_model(full_cfg, full_sd, DEV, dtype=H)
    blk = m.blocks[0]
    hs = a["h"].to(DEV).half()
    rot = (a["rot_cos"].to(DEV), a["rot_sin"].to(DEV))
    outs = {"a_self": blk.attn1(hidden_states=hs, rotary_emb=rot),
            "a_cross": blk.attn2(hidden_states=hs, encoder_hidden_states=a["ctx"].to(DEV).half())}
    for k, out in outs.items():
        r32, rh = rel_rms(out, a[k]), rel_rms(out, h[k + "_fp16"])
        record(f"fp16_wan_block_tiny[{k}]", "rel_rms hip fp16 vs reference fp32 run", r32, 4e-3)
        record(f"fp16_wan_block_tiny[{k} vs ref fp16]", "rel_rms hip fp16 vs reference fp16 run", rh, 2.5e-3)
        assert out.dtype == H and r32 < 4e-3 and rh < 2.5e-3, (k, r32, rh)


@pytest.mark.parametrize("case,modpath", [("ffn", "ffn.net.0.proj"), ("attn", "attn1.to_out.0")])
def test_wan_fp16_saturation_nan_and_finite_in_the_same_elements_as_the_reference(golden, case, modpath):
    """ONE output neuron of one linear of the LAST block scaled (in fp32, then cast -- as the generator did) until 9 of the 72 tokens
    overflow there; from that point to the output everything is per token, so the reference's fp16 run returns nan for those tokens
    and finite numbers for the others:
      ffn : a pre-activation of the FFN's first linear rounds to +-inf; gelu(+inf) = inf, gelu(-inf) = nan (transformer_wan.py:345
            through diffusers' GELU); the row leaves the second linear as inf / nan, the gated residual (:348) keeps it;
      attn: an element of attn_output (fp16) overflows, or hidden + attn_output * gate (:336) does; norm2 (:339) turns the row into nan.
    The HIP path fuses bias + GELU and the gated residual into GEMM epilogues: it must round to fp16 exactly where the reference
    does, or a token that is nan there comes out finite here (and the other way round)."""
    cfg, sd, a = golden("wan_dit_tiny")
    _, _, h = golden("wan_dit_tiny_fp16")
    gain, layer, j = float(h[f"sat_{case}_gain"]), int(h[f"sat_{case}_layer"]), int(h[f"sat_{case}_neuron"])
    sd = {k: v.clone() for k, v in sd.items()}
    for leaf in ("weight", "bias"):
        sd[f"blocks.{layer}.{modpath}.{leaf}"][j] *= gain                     # fp32 product; load_reference_state_dict casts to fp16
    m = hip_wan_model(cfg, sd, DEV, dtype=H)
    out = m(a["x"].to(DEV).half(), a["ts_tok"].to(DEV), a["txt"].to(DEV).half(), return_dict=False)[0].float().cpu()
    ref = h[f"y_tok_fp16_sat_{case}"].float()
    assert out.shape == ref.shape
    n_bad_ref = int((~torch.isfinite(ref)).sum())
    assert 0 < n_bad_ref < ref.numel()                                       # the fixture has both kinds of tokens
    bad = _classes_differ(out, ref)
    fin = torch.isfinite(ref) & torch.isfinite(out)
    r = rel_rms(out[fin], ref[fin])
    record(f"fp16_saturation[{case}]", f"elements whose inf / nan / finite class differs from the reference fp16 run "
           f"(of {ref.numel()}; reference: {n_bad_ref} non-finite; finite elements rel_rms {r:.5f})", bad, 0)
    assert bad == 0, (case, bad, n_bad_ref)
    # (the finite tokens carry the scaled neuron too -- values up to ~6e4, one fp16 ulp there is 32 -- so they sit further from the
    # reference's run than on the unscaled model: 2.7e-3 measured on the attn case against 6e-4)
    assert r < 5e-3, r


# ------------------------------------------------------------------------------------------------ CogVideoX DiT
@pytest.mark.parametrize("tag", ["def", "rsz"])
def test_cog_fp16_model_vs_the_reference_fp16_and_fp32_runs(golden, tag):
    from frameino_amd.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from tests.test_oracle_golden import _cog_cfg
    cfg, sd, a = golden("cog_dit_tiny")
    _, _, h = golden("cog_dit_tiny_fp16")
    m = CogVideoXTransformer3DModel(**_cog_cfg(cfg)).to(DEV)
    m.load_reference_state_dict(sd, dtype=H)
    m.eval()
    out = m(hidden_states=a[f"x_{tag}"].to(DEV).half(), encoder_hidden_states=a[f"txt_{tag}"].to(DEV).half(),
            timestep=a[f"ts_{tag}"].to(DEV), image_rotary_emb=(a[f"cos_{tag}"].to(DEV), a[f"sin_{tag}"].to(DEV)),
            return_dict=False)[0]
    ref32, refh = a[f"y_{tag}"], h[f"y_{tag}_fp16"]
    r32, rh, ref_h = rel_rms(out, ref32), rel_rms(out, refh), rel_rms(refh, ref32)
    record(f"fp16_cog_dit_tiny[{tag}]", f"rel_rms hip fp16 vs reference fp32 run (reference's own fp16 run: {ref_h:.5f})", r32, 5e-3)
    record(f"fp16_cog_dit_tiny[{tag} vs ref fp16]", "rel_rms hip fp16 vs reference fp16 run", rh, 4e-3)
    assert out.dtype == H and out.shape == ref32.shape and r32 < 5e-3 and rh < 4e-3, (r32, rh)


# ------------------------------------------------------------------------------------------------ the two drop-in calls
def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10 * math.log10(1.0 / max(mse, 1e-20))


@pytest.mark.parametrize("sched", ["euler", "unipc"])
def test_wan_call_in_the_apps_precision_mix_vs_the_reference_run(golden, sched):
    """app.py:156-157: DiT fp16 (fp32 islands), VAE fp32 -- here the VAE in its fp32-compute mode (`set_compute_dtype(float32)`,
    split-bf16 products) -- through `__call__`: encodes of the conditions, the loop, decode, post-processing; against the
    reference pipeline's own run in that mix (tests/golden/wan_pipe_fp16_tiny.npz) and its all-fp32 run."""
    import PIL.Image
    from frameino_amd.pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline
    from frameino_amd.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
    from tests.test_wan_vae_gpu import _vae
    cfg, sd, a = golden("wan_pipe_tiny")
    _, _, h = golden("wan_pipe_fp16_tiny")
    _, _, u = golden("wan_pipe_unipc_tiny")
    m = hip_wan_model(cfg, {k[4:]: v for k, v in sd.items() if k.startswith("dit.")}, DEV, dtype=H)
    vae, _ = _vae(golden, "wan_pipe_tiny", prefix="vae")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae.load_reference_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("vae.")}, dtype=torch.float32)   # app.py:157
    vae.set_compute_dtype(torch.float32)
    assert vae.dtype == torch.float32 and vae.compute_dtype == torch.float32
    s = FlowMatchEulerDiscreteScheduler(shift=5.0) if sched == "euler" else UniPCMultistepScheduler(flow_shift=5.0)
    pipe = WanImageToVideoPipeline(vae=vae, scheduler=s, transformer=m, expand_timesteps=True)
    img = PIL.Image.fromarray(a["image"].numpy() if hasattr(a["image"], "numpy") else a["image"])
    hh, ww = img.size[1], img.size[0]
    steps = int(a["steps"]) if sched == "euler" else int(u["steps"])
    kw = dict(image=img, prompt_embeds=a["prompt_embeds"], negative_prompt_embeds=a["negative_embeds"], traj_tensor=a["traj"],
              ID_tensor=a["id_tensor"], height=hh, width=ww, num_frames=a["traj"].shape[0], num_inference_steps=steps,
              guidance_scale=float(a["guidance"]))
    lat = pipe(latents=a["latents0"].clone(), output_type="latent", **kw).frames
    ref_h = h["out_latents_fp16dit"] if sched == "euler" else h["out_latents_unipc_fp16dit"]
    ref_32 = a["out_latents"] if sched == "euler" else u["out_latents"]
    rh, r32, rr = rel_rms(lat, ref_h), rel_rms(lat, ref_32), rel_rms(ref_h, ref_32)
    record(f"fp16_wan_call[{sched} latents]", f"rel_rms hip (fp16 DiT, fp32-compute VAE) vs reference fp32 run (reference's own "
           f"fp16-DiT run: {rr:.5f})", r32, 1e-2)
    record(f"fp16_wan_call[{sched} latents vs ref fp16-DiT run]", "rel_rms", rh, 1e-2)
    assert lat.shape == ref_32.shape and r32 < 1e-2 and rh < 1e-2, (r32, rh, rr)
    if sched == "euler":
        vid = pipe(latents=a["latents0"].clone(), output_type="np", **kw).frames
        p32, ph = _psnr(vid, a["out_video"].numpy()), _psnr(vid, h["out_video_fp16dit"].numpy())
        pref = _psnr(h["out_video_fp16dit"].numpy(), a["out_video"].numpy())
        record("fp16_wan_call[video]", f"PSNR dB hip vs reference fp32 run (reference's own fp16-DiT run: {pref:.1f} dB; higher is "
               f"better)", p32, pref - 3.0, lower_is_better=False)
        assert vid.shape == tuple(a["out_video"].shape) and p32 > pref - 3.0 and ph > pref - 3.0, (p32, ph, pref)


def test_cog_call_all_fp16_vs_the_reference_run(golden):
    """test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-94,106: transformer, VAE and embeddings in fp16.  `__call__` end to
    end (prepare_latents, trajectory / ID encodes through the HIP VAE, DDIM loop with and without dynamic CFG, DPM with the
    same generator, decode) against the reference pipeline's own all-fp16 run (tests/golden/cog_pipe_fp16_tiny.npz)."""
    import PIL.Image
    from frameino_amd.schedulers import CogVideoXDDIMScheduler, CogVideoXDPMScheduler
    from tests.test_cog_model_gpu import _cog_pipe
    _, _, h = golden("cog_pipe_fp16_tiny")
    for key, ref_key, sched, extra in (("out_ddim", "out_ddim_fp16", CogVideoXDDIMScheduler, {}),
                                       ("out_ddim_dynamic_cfg", "out_ddim_dynamic_cfg_fp16", CogVideoXDDIMScheduler, {"use_dynamic_cfg": True}),
                                       ("out_dpm", "out_dpm_fp16", CogVideoXDPMScheduler, {"generator": 11})):
        pipe, a, _ = _cog_pipe(golden, sched(), with_vae=True, dtype=H)
        hh, ww = a["image"].shape[:2]
        kw = dict(image=PIL.Image.fromarray(a["image"].numpy()), traj_tensor=a["traj"].to(DEV), ID_tensor=a["id_tensor"].to(DEV),
                  prompt_embeds=a["prompt_embeds"].to(DEV).half(), negative_prompt_embeds=a["negative_embeds"].to(DEV).half(),
                  height=hh, width=ww, num_frames=a["traj"].shape[0], num_inference_steps=int(a["steps"]),
                  guidance_scale=float(a["guidance"]), add_ID_reference_augment_noise=False, latents=a["latents0"].to(DEV).half())
        if "generator" in extra:
            extra = dict(extra, generator=torch.Generator().manual_seed(extra["generator"]))
        torch.manual_seed(7)
        lat = pipe(output_type="latent", **kw, **extra).frames
        rh = rel_rms(lat, h[ref_key])
        bound = 2e-2 if key != "out_dpm" else 4e-2
        record(f"fp16_cog_call[{key}]", "rel_rms hip all-fp16 __call__ latents vs the reference's all-fp16 run", rh, bound)
        assert lat.shape == h[ref_key].shape and torch.isfinite(lat.float()).all() and rh < bound, (key, rh)
        if key == "out_ddim":
            r32 = rel_rms(lat, a["out_ddim"])
            rr = rel_rms(h[ref_key], a["out_ddim"])
            record("fp16_cog_call[out_ddim vs fp32 run]", f"rel_rms (reference's own fp16 run vs its fp32 run: {rr:.5f})", r32, 2e-2)
            assert r32 < 2e-2, (r32, rr)
            torch.manual_seed(7)
            vid = pipe(output_type="np", **kw).frames
            p32, pref = _psnr(vid, a["out_video"].numpy()), _psnr(h["out_video_fp16"].numpy(), a["out_video"].numpy())
            record("fp16_cog_call[video]", f"PSNR dB hip all-fp16 video vs reference fp32 run (reference's own fp16 run: {pref:.1f} dB; "
                   f"higher is better)", p32, pref - 3.0, lower_is_better=False)
            assert p32 > pref - 3.0, (p32, pref)


# ------------------------------------------------------------------------------------------------ kernel level, overflow included
@pytest.mark.parametrize("epi", ["bias", "gelu", "residual", "gated"])
def test_gemm_epilogues_fp16_round_where_torch_fp16_rounds_overflow_included(epi):
    """y = T(x W^T + b) is the reference's fp16 linear; GELU / residual / gate come after that rounding.  Inputs sized so that ~1 %
    of the linear's outputs exceed 65504: +-inf after the linear, then gelu(-inf) = nan, inf * gate, inf + residual ..."""
    from frameino_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    mm, n, k = 384, 512, 256
    x = (torch.randn(mm, k, device=DEV, generator=g) * 60).half()
    w = (torch.randn(n, k, device=DEV, generator=g) * 30).half()        # sigma of the product sum = 60 * 30 * 16 = 28800: 2.3 % beyond 65504
    b = torch.randn(n, device=DEV, generator=g).half()
    lin = torch.nn.functional.linear(x.float(), w.float(), b.float()).half()                # fp32 accumulate, ONE rounding to fp16
    assert 0 < int(torch.isinf(lin).sum()) < lin.numel() // 10
    res = (torch.randn(mm, n, device=DEV, generator=g) * 100).half()
    gate = torch.randn(2, n, device=DEV, generator=g)
    gate[0, :8] = 0.0                                                    # inf * 0 = nan
    sel = (torch.arange(mm, device=DEV) % 2).to(torch.int32)
    if epi == "bias":
        out, ref = ops.gemm(x, w, b), lin
    elif epi == "gelu":
        out = ops.gemm(x, w, b, ops.EPI_GELU_TANH)
        ref = torch.nn.functional.gelu(lin.float(), approximate="tanh").half()
    elif epi == "residual":
        out = ops.gemm(x, w, b, ops.EPI_RESIDUAL, residual=res)
        ref = (res.float() + lin.float()).half()
    else:
        out = ops.gemm(x, w, b, ops.EPI_GATED_RESIDUAL, residual=res, gate=gate, sel=sel)
        ref = (res.float() + lin.float() * gate[sel.long()]).half()      # (hidden.float() + y * gate).type_as(hidden)  :336
    bad = _classes_differ(out, ref)
    fin = torch.isfinite(ref) & torch.isfinite(out)
    r = rel_rms(out[fin], ref[fin])
    record(f"fp16_gemm_epilogue_overflow[{epi}]", f"elements whose inf / nan / finite class differs from torch fp16 (finite rel_rms {r:.2e})",
           bad, 0)
    assert out.dtype == H and bad == 0 and r < 1e-3, (epi, bad, r)


def test_cfg_euler_chain_fp16_model_dtype_rounding():
    """noise_pred = uncond + g * (cond - uncond) in the model dtype (:882), the scheduler step in fp32: T(u + T(g . T(c - u))) with
    T = fp16, large predictions included (g . (c - u) beyond 65504 -> inf in the reference's fp16 arithmetic as well)."""
    from frameino_amd import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    lat = torch.randn(4, 3, 8, 12, device=DEV, generator=g)
    c = (torch.randn(4, 3, 8, 12, device=DEV, generator=g) * 3).half()
    u = (torch.randn(4, 3, 8, 12, device=DEV, generator=g) * 3).half()
    c.view(-1)[:16] *= 9000
    dt = torch.tensor([-0.05], device=DEV)
    exp_pred = (u + (5.0 * (c - u)))                                     # torch fp16 ops: each rounds to fp16
    exp = lat + dt * exp_pred.float()
    out = lat.clone()
    ops.cfg_euler_step_(out, c, u, 5.0, dt, round_out=False)
    assert _classes_differ(out, exp) == 0
    fin = torch.isfinite(exp)
    assert int((~fin).sum()) > 0
    torch.testing.assert_close(out[fin], exp[fin], atol=2e-6, rtol=1e-6)


def test_text_out_projection_reassociated_in_fp16_and_its_fallback_when_it_leaves_the_range():
    """P.(V W_o^T) stores V W_o^T in fp16.  (a) at ordinary magnitudes the re-associated forward matches the (P.V) W_o^T forward and
    the fp32 oracle within the fp16 bound; (b) with attn2's to_v and to_out scaled until elements of V W_o^T exceed 65504 -- while the
    reference's order stays finite: P averages ~65 keys with random signs before W_o multiplies -- the model must drop the
    re-association for that prompt (`w2 is None`) and still match the oracle."""
    from oracle import wan_dit as W
    cfg = dict(W.WAN22_5B_CFG, num_attention_heads=4, attention_head_dim=128, in_channels=16, out_channels=8,
               text_dim=256, ffn_dim=1024, num_layers=2)
    sd = W.wan_random_state_dict(cfg, seed=7, dtype=torch.float32, std=0.04)
    g = torch.Generator().manual_seed(18)
    x = torch.randn(1, 16, 5, 16, 20, generator=g)
    txt = torch.randn(1, 512, 256, generator=g)
    txt[0, 64:] = 0
    ts = torch.tensor([811.0])
    xd, td, tsd = x.to(DEV).half(), txt.to(DEV).half(), ts.to(DEV)
    # (a)
    ref32 = W.wan_forward(sd, cfg, x, ts, txt)
    m = hip_wan_model(cfg, sd, DEV, dtype=H)
    re = m(xd, tsd, td, return_dict=False)[0]
    hit = next(iter(m._text_cache.values()))[2]
    assert hit.w2 is not None
    mx = max(float(w.abs().max()) for layer in hit.w2 for w in layer)
    m.reassociate_text_out = False
    pv = m(xd, tsd, td, return_dict=False)[0]
    r_re, r_pv, r_both = rel_rms(re, ref32), rel_rms(pv, ref32), rel_rms(re, pv)
    record("fp16_text_out_reassociated", f"rel_rms P.(V Wo^T) forward vs oracle fp32 ((P.V) Wo^T forward: {r_pv:.5f}; the two "
           f"forwards: {r_both:.5f})", r_re, 4e-3)
    assert r_re < 4e-3 and r_re < 1.3 * r_pv + 3e-4 and r_both < 2e-3, (r_re, r_pv, r_both)
    # (b)
    total = 2.0 ** math.ceil(math.log2(1.5 * 65504.0 / mx))
    sv = 2.0 ** (int(math.log2(total)) // 2)
    so = total / sv
    sds = dict(sd)
    for i in range(cfg["num_layers"]):
        for leaf in ("weight", "bias"):
            sds[f"blocks.{i}.attn2.to_v.{leaf}"] = sd[f"blocks.{i}.attn2.to_v.{leaf}"] * sv
        sds[f"blocks.{i}.attn2.to_out.0.weight"] = sd[f"blocks.{i}.attn2.to_out.0.weight"] * so
    ref32s = W.wan_forward(sds, cfg, x, ts, txt)
    assert torch.isfinite(ref32s).all()
    ms = hip_wan_model(cfg, sds, DEV, dtype=H)
    out = ms(xd, tsd, td, return_dict=False)[0]
    hit = next(iter(ms._text_cache.values()))[2]
    assert hit.tail is not None and hit.w2 is None, "V W_o^T left fp16 range: the re-association must be dropped for this prompt"
    r_s = rel_rms(out, ref32s)
    record("fp16_text_out_reassociated[fallback]", f"rel_rms vs oracle fp32 with V Wo^T beyond fp16 range (scale {total:g})", r_s, 4e-3)
    assert torch.isfinite(out.float()).all() and r_s < 4e-3, r_s


# ------------------------------------------------------------------------------------------------ CogVideoX VAE in fp16
@pytest.mark.parametrize("frames", [1, 9, 17])
def test_cog_vae_fp16_encode_decode_vs_oracle(frames):
    """The evaluation script loads AutoencoderKLCogVideoX in fp16 (`:94`).  HIP fp16 vs the oracle's restatement in fp32 on the same
    fp16-rounded weights (third-party: parity unpinned -- what is checked is HIP == restatement) and vs the oracle run in fp16."""
    from frameino_amd.autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
    from oracle import cog_vae as V
    from tests.test_cog_vae_gpu import TINY
    sd = {k: v.half().float() for k, v in V.cog_vae_random_state_dict(TINY, 3).items()}
    vae = AutoencoderKLCogVideoX(**TINY).to(DEV)
    vae.load_reference_state_dict(sd, dtype=H)
    assert vae.dtype == H
    g = torch.Generator().manual_seed(40 + frames)
    x = (torch.rand(1, 3, frames, 32, 48, generator=g) * 2 - 1).half()
    ref = V.encode_moments(sd, TINY, x.float())
    out = vae.encode(x.to(DEV)).latent_dist.parameters
    r = rel_rms(out, ref)
    record(f"fp16_cog_vae[encode {frames} frames]", "rel_rms hip fp16 vs oracle fp32 (bf16 measures ~1.3e-2)", r, 4e-3)
    assert out.dtype == H and out.shape == ref.shape and r < 4e-3, r
    z = torch.randn(1, 4, 1 + (frames - 1) // 4, 4, 6, generator=g).half()
    refd = V.decode(sd, TINY, z.float())
    outd = vae.decode(z.to(DEV)).sample
    rd = rel_rms(outd, refd)
    record(f"fp16_cog_vae[decode {z.shape[2]} latent frames]", "rel_rms hip fp16 vs oracle fp32", rd, 5e-3)
    assert outd.dtype == H and outd.shape == refd.shape and rd < 5e-3, rd
