#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own model files on CPU.

Runs only in the build container (needs /root/reference).  The reference is imported
unmodified, through the builder-written `diffusers` stand-in in ./diffusers_stub (diffusers
is not installable offline).  Every fixture stores: the config, the seeded random weights
(`sd/<reference parameter name>`), the inputs and the reference's outputs.

    python tools/golden/make_golden.py [--only wan_dit,wan_pipe,...]

Fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("FINO_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(HERE, "diffusers_stub"))
sys.path.insert(0, "/root/reference")
os.chdir("/root/reference")  # the reference does sys.path.append(os.path.abspath('.'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)


def to_np(t):
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            return t.float().numpy()
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, cfg=None, sd=None, **arrays):
    d = {}
    if cfg is not None:
        for k, v in cfg.items():
            d["cfg/" + k] = np.asarray(v if v is not None else -1)
    if sd is not None:
        for k, v in sd.items():
            d["sd/" + k] = to_np(v)
    for k, v in arrays.items():
        d[k] = to_np(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def randomize_(module, seed, std=0.2):
    """Re-initialise every parameter with seeded noise (default inits leave e.g. norm gains at 1 and
    would hide missing-multiply bugs)."""
    g = torch.Generator().manual_seed(seed)
    for n, p in module.named_parameters():
        if n.endswith("norm_q.weight") or n.endswith("norm_k.weight") or ("norm" in n and n.endswith("weight")) \
                or n.endswith("gamma"):
            p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
        elif "scale_shift_table" in n:
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        else:
            p.copy_(std * torch.randn(p.shape, generator=g))


# ----------------------------------------------------------------------------------- Wan DiT
WAN_TINY = dict(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=8, out_channels=4,
                text_dim=16, freq_dim=32, ffn_dim=64, num_layers=2, cross_attn_norm=True, eps=1e-6,
                rope_max_seq_len=64)


def gen_wan_dit():
    from architecture.transformer_wan import (WanAttnProcessor2_0, WanRotaryPosEmbed, WanTransformer3DModel,
                                              WanTransformerBlock)

    # G3: rope tables at the real 704x1280 / 14-frame geometry, sampled rows
    rope = WanRotaryPosEmbed(128, (1, 2, 2), 1024)
    cos, sin = rope(torch.zeros(1, 1, 14, 44, 80))
    rows = torch.tensor([0, 1, 39, 40, 879, 880, 881, 5000, 12319])
    save("wan_rope", rows=rows, cos=cos[0, 0, rows], sin=sin[0, 0, rows], shape=np.array([14, 22, 40, 128]))

    # G4: tiny full model, scalar / per-token timestep
    torch.manual_seed(0)
    m = WanTransformer3DModel(**{k: v for k, v in WAN_TINY.items()}).eval()
    randomize_(m, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 8, 3, 8, 12, generator=g)
    txt = torch.randn(1, 20, 16, generator=g)
    L = 3 * 4 * 6
    ts_scalar = torch.tensor([437.0])
    ts_tok = torch.full((1, L), 437.0)
    ts_tok[0, : 4 * 6] = 0.0                      # first-frame tokens see t=0 (pipeline :832-843)
    ts_many = torch.rand(1, L, generator=g) * 1000  # general per-token values
    y_scalar = m(x, ts_scalar, txt, return_dict=False)[0]
    y_tok = m(x, ts_tok, txt, return_dict=False)[0]
    y_many = m(x, ts_many, txt, return_dict=False)[0]
    save("wan_dit_tiny", cfg=WAN_TINY, sd=m.state_dict(), x=x, txt=txt, ts_scalar=ts_scalar, ts_tok=ts_tok,
         ts_many=ts_many, y_scalar=y_scalar, y_tok=y_tok, y_many=y_many)

    # G1/G2: one block + its two attention calls in isolation
    blk = m.blocks[0]
    h = torch.randn(1, L, 256, generator=g)
    ctx = torch.randn(1, 20, 256, generator=g)
    rot = m.rope(x)
    temb4 = torch.randn(1, L, 6, 256, generator=g) * 0.3
    temb3 = torch.randn(1, 6, 256, generator=g) * 0.3
    a_self = blk.attn1(hidden_states=h, rotary_emb=rot)
    a_cross = blk.attn2(hidden_states=h, encoder_hidden_states=ctx)
    b4 = blk(h, ctx, temb4, rot)
    b3 = blk(h, ctx, temb3, rot)
    sd_blk = {"blocks.0." + k: v for k, v in blk.state_dict().items()}
    save("wan_block_tiny", cfg=WAN_TINY, sd=sd_blk, h=h, ctx=ctx, rot_cos=rot[0], rot_sin=rot[1], temb4=temb4,
         temb3=temb3, a_self=a_self, a_cross=a_cross, b4=b4, b3=b3)

    # bf16 variant of G4 (fixes the stated tolerance).  fp32 islands per _keep_in_fp32_modules (:393)
    mb = WanTransformer3DModel(**{k: v for k, v in WAN_TINY.items()}).eval()
    mb.load_state_dict(m.state_dict())
    keep = WanTransformer3DModel._keep_in_fp32_modules
    for n, p in mb.named_parameters():
        if not any(k in n for k in keep):
            p.data = p.data.to(torch.bfloat16)
    yb = mb(x.bfloat16(), ts_tok, txt.bfloat16(), return_dict=False)[0]
    save("wan_dit_tiny_bf16", y_tok_bf16=yb)

    # round 6: the same model the way the canonical caller loads it -- app.py:156 `torch_dtype=torch.float16` (from_pretrained
    # keeps `_keep_in_fp32_modules` in fp32 for any half dtype) -- scalar / {0,t} / general per-token timesteps, and the
    # two attention calls + the block in isolation.  Outputs only: weights and inputs are wan_dit_tiny's / wan_block_tiny's.
    mh = WanTransformer3DModel(**{k: v for k, v in WAN_TINY.items()}).eval()
    mh.load_state_dict(m.state_dict())
    for n, p in mh.named_parameters():
        if not any(k in n for k in keep):
            p.data = p.data.to(torch.float16)
    xh, th = x.half(), txt.half()
    bh = mh.blocks[0]
    hh, ch = h.half(), ctx.half()
    # SATURATION (fp16 ends at 65504): which elements of the reference's own fp16 run come out inf / nan when an intermediate
    # leaves the range is part of its behaviour at this dtype.  Two model-level cases on wan_dit_tiny's inputs (ts_tok), each with
    # ONE weight of the LAST block scaled (in fp32, then cast) so that only a handful of intermediate elements overflow; from there
    # to the output everything is per token (cross-attention queries, FFN, norm_out, proj_out), so the other tokens stay finite:
    #   ffn : blocks.1.ffn.net.0.proj x g -> a few outputs of the FFN's first linear round to +-inf; gelu(+inf) = inf, gelu(-inf) = nan
    #         (:345); those ROWS leave the second linear as inf / nan and the gated residual (:348) keeps them
    #   attn: blocks.1.attn1.to_out.0 x g -> a few elements of attn_output (fp16) or of hidden + attn_output * gate (:336) overflow;
    #         norm2 (:339) turns such a row into nan
    import copy
    sat = {}
    with torch.no_grad():
        last = len(m.blocks) - 1
        pre = {}
        hk1 = m.blocks[last].ffn.net[0].proj.register_forward_hook(lambda mod, i, o: pre.__setitem__("ffn", o.detach()))
        hk2 = m.blocks[last].attn1.to_out[0].register_forward_hook(lambda mod, i, o: pre.__setitem__("attn", o.detach()))
        m(x, ts_tok, txt, return_dict=False)                       # the fp32 model's own pre-activations set the gains
        hk1.remove(); hk2.remove()
        for name, modpath, nth in (("ffn", f"blocks.{last}.ffn.net.0.proj", 8), ("attn", f"blocks.{last}.attn1.to_out.0", 8)):
            # ONE output neuron j of that linear is scaled (row j of the weight and bias[j]): the one whose |pre-activation|,
            # sorted over the tokens, has the widest gap after the 9th token -- the threshold sits in the middle of that gap, so
            # that fp16 rounding upstream cannot move a token across it, and 9 of the 72 tokens overflow
            v = pre[name].abs().flatten(0, -2).sort(dim=0, descending=True).values      # [tokens, neurons]
            j = int((v[nth] / v[nth + 1]).argmax())
            gain = float(65520.0 / (v[nth, j] * v[nth + 1, j]).sqrt())
            ms = copy.deepcopy(mh)
            lin, lin32 = ms.get_submodule(modpath), m.get_submodule(modpath)
            lin.weight[j].copy_((lin32.weight[j] * gain).half())
            lin.bias[j].copy_((lin32.bias[j] * gain).half())
            sat[f"sat_{name}_neuron"] = np.array(j)
            print(f"  neuron {j}: gap {float(v[nth, j] / v[nth + 1, j]):.3f}")
            out = ms(xh, ts_tok, th, return_dict=False)[0]
            sat[f"y_tok_fp16_sat_{name}"] = out
            sat[f"sat_{name}_gain"] = np.array(gain)
            sat[f"sat_{name}_layer"] = np.array(last)
            print(f"  saturation case {name}: gain {gain:g}; inf {int(torch.isinf(out).sum())}, nan {int(torch.isnan(out).sum())} of {out.numel()}")
    save("wan_dit_tiny_fp16", y_scalar_fp16=mh(xh, ts_scalar, th, return_dict=False)[0],
         y_tok_fp16=mh(xh, ts_tok, th, return_dict=False)[0], y_many_fp16=mh(xh, ts_many, th, return_dict=False)[0],
         a_self_fp16=bh.attn1(hidden_states=hh, rotary_emb=rot), a_cross_fp16=bh.attn2(hidden_states=hh, encoder_hidden_states=ch),
         b4_fp16=bh(hh, ch, temb4, rot), b3_fp16=bh(hh, ch, temb3, rot), **sat)


# ----------------------------------------------------------------------------------- Wan pipeline
def _ftfy_stand_in():
    """`ftfy` is absent offline and the reference's prompt_clean (pipelines/pipeline_wan_i2v_motion_FrameINO.py:103-117)
    calls ftfy.fix_text whenever a prompt STRING is given.  For the plain-ASCII prompts recorded here fix_text is the
    identity (it repairs mojibake), so an identity module stands in; html.unescape and the whitespace collapse that follow
    are the reference's own code and are exercised by the prompt below."""
    import types
    if "ftfy" not in sys.modules:
        mod = types.ModuleType("ftfy")
        mod.fix_text = lambda s: s
        sys.modules["ftfy"] = mod


WAN_PROMPT = "  A red ball   rolls to the right &amp;amp; stops.\n"      # doubled spaces, a double-escaped entity, a newline
COG_PROMPT = "a red ball"


def gen_wan_pipe():
    import PIL.Image
    _ftfy_stand_in()
    from diffusers.schedulers import FlowMatchEulerDiscreteScheduler, UniPCMultistepScheduler
    from architecture.autoencoder_kl_wan import AutoencoderKLWan
    from architecture.transformer_wan import WanTransformer3DModel
    from pipelines.pipeline_wan_i2v_motion_FrameINO import WanImageToVideoPipeline

    zdim = 4
    vae_cfg = dict(base_dim=8, decoder_base_dim=8, z_dim=zdim, dim_mult=[1, 2, 4, 4], num_res_blocks=1,
                   attn_scales=[], temperal_downsample=[False, True, True], dropout=0.0,
                   latents_mean=[0.1, -0.2, 0.3, 0.05], latents_std=[1.1, 0.9, 1.3, 0.7], is_residual=True,
                   in_channels=12, out_channels=12, patch_size=2, scale_factor_temporal=4, scale_factor_spatial=16)
    torch.manual_seed(0)
    vae = AutoencoderKLWan(**vae_cfg).eval()
    randomize_(vae, 11, std=0.15)
    dit_cfg = dict(WAN_TINY, in_channels=2 * zdim, out_channels=zdim)
    dit = WanTransformer3DModel(**dit_cfg).eval()
    randomize_(dit, 12)
    sched = FlowMatchEulerDiscreteScheduler(shift=5.0)
    pipe = WanImageToVideoPipeline(tokenizer=None, text_encoder=None, vae=vae, scheduler=sched, transformer=dit,
                                   expand_timesteps=True)

    H, W, F = 64, 96, 9                      # latent 4x6, 3 latent frames (+1 ID)
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy()
    traj = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    idt = torch.rand(1, 3, 1, H, W, generator=g) * 2 - 1
    pe = torch.randn(1, 12, 16, generator=g)
    ne = torch.randn(1, 12, 16, generator=g)
    lat0 = torch.randn(1, zdim, 3, H // 16, W // 16, generator=g)

    rec = {}
    orig = pipe.prepare_latents

    def spy(*a, **k):
        out = orig(*a, **k)
        rec["latents"], rec["condition"], rec["traj_latents"], rec["ID_latent"], rec["mask"] = out
        return out

    pipe.prepare_latents = spy
    steps = 4
    out_lat = pipe(image=PIL.Image.fromarray(img), prompt_embeds=pe, negative_prompt_embeds=ne, traj_tensor=traj,
                   ID_tensor=idt, height=H, width=W, num_frames=F, num_inference_steps=steps, guidance_scale=5.0,
                   latents=lat0.clone(), output_type="latent").frames
    out_np = pipe(image=PIL.Image.fromarray(img), prompt_embeds=pe, negative_prompt_embeds=ne, traj_tensor=traj,
                  ID_tensor=idt, height=H, width=W, num_frames=F, num_inference_steps=steps, guidance_scale=5.0,
                  latents=lat0.clone(), output_type="np").frames
    # the same call driven by the UniPC scheduler the released Wan2.2 folder ships (stand-in scheduler, reference loop):
    # 6 steps = order-1 start, order-2 middle, order-1 final step, corrector throughout
    pipe.scheduler = UniPCMultistepScheduler(flow_shift=5.0)
    out_unipc = pipe(image=PIL.Image.fromarray(img), prompt_embeds=pe, negative_prompt_embeds=ne, traj_tensor=traj,
                     ID_tensor=idt, height=H, width=W, num_frames=F, num_inference_steps=6, guidance_scale=5.0,
                     latents=lat0.clone(), output_type="latent").frames
    np.savez_compressed(os.path.join(OUT, "wan_pipe_unipc_tiny.npz"), out_latents=to_np(out_unipc), steps=np.array(6),
                        timesteps=to_np(pipe.scheduler.timesteps))
    print("wrote wan_pipe_unipc_tiny.npz (outputs only: weights and inputs are wan_pipe_tiny's)")
    # the `prompt=` route both canonical callers take (app.py:708-719): the reference's own encode_prompt /
    # _get_t5_prompt_embeds (:206-337: prompt_clean, tokenizer call, UMT5 encoder, zero padding past the true length) on a
    # toy UMT5EncoderModel (the real transformers class) and a character tokenizer (tests/text_stub.py)
    sys.path.insert(0, REPO)
    from tests.text_stub import CharTokenizer, tiny_text_encoder
    te = tiny_text_encoder("umt5", seed=71)
    pipe_t = WanImageToVideoPipeline(tokenizer=CharTokenizer(), text_encoder=te, vae=vae,
                                     scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0), transformer=dit,
                                     expand_timesteps=True)
    pe_t, ne_t = pipe_t.encode_prompt(WAN_PROMPT, "", True, 1, max_sequence_length=512)
    out_prompt = pipe_t(image=PIL.Image.fromarray(img), prompt=WAN_PROMPT, negative_prompt="", traj_tensor=traj,
                        ID_tensor=idt, height=H, width=W, num_frames=F, num_inference_steps=steps, guidance_scale=5.0,
                        latents=lat0.clone(), output_type="latent").frames
    text = {"te/" + k: v for k, v in te.state_dict().items()}
    text.update(prompt=np.array(WAN_PROMPT), negative_prompt=np.array(""), prompt_embeds_from_text=pe_t,
                negative_embeds_from_text=ne_t, out_latents_prompt=out_prompt)
    # round 6: the app's precision mix -- app.py:156-157: the DiT in fp16 (fp32 islands kept), the VAE in fp32; the pipeline
    # casts prompt embeddings to the DiT's dtype (:781-783) -- Euler and UniPC-driven.  Outputs only (wan_pipe_tiny's weights).
    import copy
    dit_h = copy.deepcopy(dit)
    keep = WanTransformer3DModel._keep_in_fp32_modules
    for n, p in dit_h.named_parameters():
        if not any(k in n for k in keep):
            p.data = p.data.to(torch.float16)
    pipe_h = WanImageToVideoPipeline(tokenizer=None, text_encoder=None, vae=vae, scheduler=FlowMatchEulerDiscreteScheduler(shift=5.0),
                                     transformer=dit_h, expand_timesteps=True)
    kw_h = dict(image=PIL.Image.fromarray(img), prompt_embeds=pe, negative_prompt_embeds=ne, traj_tensor=traj, ID_tensor=idt,
                height=H, width=W, num_frames=F, guidance_scale=5.0)
    lat_h = pipe_h(**kw_h, num_inference_steps=steps, latents=lat0.clone(), output_type="latent").frames
    vid_h = pipe_h(**kw_h, num_inference_steps=steps, latents=lat0.clone(), output_type="np").frames
    pipe_h.scheduler = UniPCMultistepScheduler(flow_shift=5.0)
    lat_hu = pipe_h(**kw_h, num_inference_steps=6, latents=lat0.clone(), output_type="latent").frames
    np.savez_compressed(os.path.join(OUT, "wan_pipe_fp16_tiny.npz"), out_latents_fp16dit=to_np(lat_h), out_video_fp16dit=to_np(vid_h),
                        out_latents_unipc_fp16dit=to_np(lat_hu))
    print("wrote wan_pipe_fp16_tiny.npz (outputs only: weights and inputs are wan_pipe_tiny's)")
    sched.set_timesteps(steps)
    sd = {"dit." + k: v for k, v in dit.state_dict().items()}
    sd.update({"vae." + k: v for k, v in vae.state_dict().items()})
    cfg = dict(dit_cfg)
    cfg.update({"vae_" + k: v for k, v in vae_cfg.items()})
    save("wan_pipe_tiny", cfg=cfg, sd=sd, image=img, traj=traj, id_tensor=idt, prompt_embeds=pe, negative_embeds=ne,
         latents0=lat0, condition=rec["condition"], traj_latents=rec["traj_latents"], id_latent=rec["ID_latent"],
         mask=rec["mask"], timesteps=sched.timesteps, sigmas=sched.sigmas, out_latents=out_lat, out_video=out_np,
         steps=np.array(steps), guidance=np.array(5.0), **text)


# ----------------------------------------------------------------------------------- Wan VAE
VAE_TINY = dict(base_dim=8, decoder_base_dim=16, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                temperal_downsample=[False, True, True], dropout=0.0, latents_mean=[0.1, -0.2, 0.3, 0.05],
                latents_std=[1.1, 0.9, 1.3, 0.7], is_residual=True, in_channels=12, out_channels=12, patch_size=2,
                scale_factor_temporal=4, scale_factor_spatial=16)


def gen_wan_vae():
    """G6: the reference's chunked (feat_cache streaming) encode / decode on a tiny Wan2.2-style residual VAE."""
    from architecture.autoencoder_kl_wan import AutoencoderKLWan
    torch.manual_seed(0)
    vae = AutoencoderKLWan(**VAE_TINY).eval()
    randomize_(vae, 21, std=0.12)
    g = torch.Generator().manual_seed(22)
    arrays = {}
    for nf in (1, 5, 9):
        x = torch.rand(1, 3, nf, 32, 48, generator=g) * 2 - 1
        arrays[f"enc_in_{nf}"] = x
        arrays[f"enc_out_{nf}"] = vae.encode(x).latent_dist.parameters       # moments [1, 2z, T', 2, 3]
    for nl in (1, 2, 3):
        z = torch.randn(1, 4, nl, 2, 3, generator=g)
        arrays[f"dec_in_{nl}"] = z
        arrays[f"dec_out_{nl}"] = vae.decode(z, return_dict=False)[0]
    save("wan_vae_tiny", cfg={k: v for k, v in VAE_TINY.items() if k not in ("attn_scales",)}, sd=vae.state_dict(),
         **arrays)


# ----------------------------------------------------------------------------------- CogVideoX DiT (FrameIn)
COG_TINY = dict(num_attention_heads=2, attention_head_dim=64, in_channels=6, out_channels=2, flip_sin_to_cos=True,
                freq_shift=0, time_embed_dim=32, text_embed_dim=16, num_layers=2, sample_width=8, sample_height=8,
                sample_frames=9, patch_size=2, temporal_compression_ratio=4, max_text_seq_length=8,
                norm_elementwise_affine=True, norm_eps=1e-5, use_rotary_positional_embeddings=True,
                use_learned_positional_embeddings=True, use_FrameIn=True)


def gen_cog_dit():
    """G7-G9: tiny CogVideoX FrameIn transformer (B=2, RoPE extended by the first frame's rows as the pipeline does
    :834-839), at the default resolution and at a resized one (trilinear PE resize), fused == unfused processors."""
    from architecture.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from architecture.embeddings import get_3d_rotary_pos_embed
    torch.manual_seed(0)
    m = CogVideoXTransformer3DModel(**COG_TINY).eval()
    randomize_(m, 31, std=0.15)
    with torch.no_grad():
        m.patch_embed.pos_embedding.copy_(torch.randn(m.patch_embed.pos_embedding.shape,
                                                      generator=torch.Generator().manual_seed(32)) * 0.3)
    g = torch.Generator().manual_seed(33)
    arrays = {}
    for tag, (hh, ww) in (("def", (8, 8)), ("rsz", (8, 12))):
        x = torch.randn(2, 4, 6, hh, ww, generator=g)
        txt = torch.randn(2, 8, 16, generator=g)
        ts = torch.tensor([601.0, 601.0])
        cos, sin = get_3d_rotary_pos_embed(64, ((0, 0), (hh // 2, ww // 2)), (hh // 2, ww // 2), 3)
        n1 = cos.shape[0] // 3
        cos = torch.cat([cos, cos[:n1]], dim=0)
        sin = torch.cat([sin, sin[:n1]], dim=0)
        y = m(hidden_states=x, encoder_hidden_states=txt, timestep=ts, image_rotary_emb=(cos, sin),
              return_dict=False)[0]
        arrays.update({f"x_{tag}": x, f"txt_{tag}": txt, f"ts_{tag}": ts, f"cos_{tag}": cos, f"sin_{tag}": sin,
                       f"y_{tag}": y})
    m.fuse_qkv_projections()
    yf = m(hidden_states=arrays["x_def"], encoder_hidden_states=arrays["txt_def"], timestep=arrays["ts_def"],
           image_rotary_emb=(arrays["cos_def"], arrays["sin_def"]), return_dict=False)[0]
    m.unfuse_qkv_projections()
    arrays["y_def_fused"] = yf
    sd = {k: v for k, v in m.state_dict().items() if "to_qkv" not in k}
    save("cog_dit_tiny", cfg=COG_TINY, sd=sd, **arrays)
    # round 6: all-fp16, the way test_code/run_cogvideox_FrameIn_mass_evaluation.py:92 loads it (this model has no fp32 islands:
    # no _keep_in_fp32_modules in cogvideox_transformer_3d.py).  Outputs only.
    import copy
    mh = copy.deepcopy(m).to(torch.float16)
    outs = {}
    for tag in ("def", "rsz"):
        outs[f"y_{tag}_fp16"] = mh(hidden_states=arrays[f"x_{tag}"].half(), encoder_hidden_states=arrays[f"txt_{tag}"].half(),
                                   timestep=arrays[f"ts_{tag}"], image_rotary_emb=(arrays[f"cos_{tag}"], arrays[f"sin_{tag}"]),
                                   return_dict=False)[0]
    save("cog_dit_tiny_fp16", **outs)


def gen_cog_dit_s1():
    """Stage-1 CogVideoX motion model (pipelines/pipeline_cogvideox_i2v_motion.py: `use_FrameIn=False`, no ID frame,
    RoPE not extended) -- BASELINE config 1's shape class -- at the default and a resized resolution."""
    from architecture.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from architecture.embeddings import get_3d_rotary_pos_embed
    cfg = dict(COG_TINY, use_FrameIn=False)
    torch.manual_seed(0)
    m = CogVideoXTransformer3DModel(**cfg).eval()
    randomize_(m, 51, std=0.15)
    with torch.no_grad():
        m.patch_embed.pos_embedding.copy_(torch.randn(m.patch_embed.pos_embedding.shape,
                                                      generator=torch.Generator().manual_seed(52)) * 0.3)
    g = torch.Generator().manual_seed(53)
    arrays = {}
    for tag, (hh, ww) in (("def", (8, 8)), ("rsz", (8, 12))):
        x = torch.randn(2, 3, 6, hh, ww, generator=g)
        txt = torch.randn(2, 8, 16, generator=g)
        ts = torch.tensor([401.0, 401.0])
        cos, sin = get_3d_rotary_pos_embed(64, ((0, 0), (hh // 2, ww // 2)), (hh // 2, ww // 2), 3)
        y = m(hidden_states=x, encoder_hidden_states=txt, timestep=ts, image_rotary_emb=(cos, sin),
              return_dict=False)[0]
        arrays.update({f"x_{tag}": x, f"txt_{tag}": txt, f"ts_{tag}": ts, f"cos_{tag}": cos, f"sin_{tag}": sin,
                       f"y_{tag}": y})
    save("cog_dit_s1_tiny", cfg=cfg, sd=dict(m.state_dict()), **arrays)


def gen_traj_kernel():
    """The blur kernel of the trajectory-video builder (data_loader/video_dataset_motion.py:29) from the reference's own
    function (utils/optical_flow_utils.py:197-219).  The builder itself imports cv2 (absent offline) and cannot run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ofu", "/root/reference/utils/optical_flow_utils.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    k = m.bivariate_Gaussian(45, 3, 3, 0, grid=None, isotropic=True)
    np.savez_compressed(os.path.join(OUT, "traj_kernel.npz"), kernel=k)
    print("wrote traj_kernel.npz", k.shape, k.dtype)


class _PlaceholderFinder:
    """Build container only: the reference's CogVideoX pipeline imports its TRAINING script at call time
    (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:816) for one 30-line helper, and that script's module level pulls
    cv2 / imageio / omegaconf / torchvision / wandb / ... (train_code/train_cogvideox_motion_FrameINO.py:16-60) and
    diffusers sub-modules the stand-in has no reason to have.  None of them is used by `img_tensor_to_vae_latent`: serve
    them as EMPTY placeholder modules (any attribute = an inert callable class) so that the reference file imports
    unmodified.  Modules that exist are never touched (the finder sits at the END of sys.meta_path)."""
    TOPS = {"cv2", "imageio", "omegaconf", "torchvision", "wandb", "ffmpeg", "decord", "moviepy", "av", "peft",
            "bitsandbytes", "deepspeed", "prodigyopt", "diffusers", "skimage", "scipy", "matplotlib", "pandas", "einops"}
    served = []

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] not in self.TOPS:
            return None
        self.served.append(name)
        return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        import types

        class _Inert:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return self

            def __getattr__(self, k):
                if k.startswith("__"):
                    raise AttributeError(k)
                return _Inert()

        class _Mod(types.ModuleType):
            def __getattr__(self, k):
                if k.startswith("__"):
                    raise AttributeError(k)
                return _Inert()

        m = _Mod(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


COG_PIPE_DIT = dict(num_attention_heads=2, attention_head_dim=64, in_channels=48, out_channels=16, flip_sin_to_cos=True,
                    freq_shift=0, time_embed_dim=32, text_embed_dim=16, num_layers=2, sample_width=8, sample_height=8,
                    sample_frames=9, patch_size=2, temporal_compression_ratio=4, max_text_seq_length=8,
                    norm_elementwise_affine=True, norm_eps=1e-5, use_rotary_positional_embeddings=True,
                    use_learned_positional_embeddings=True, use_FrameIn=True)
COG_PIPE_VAE = dict(in_channels=3, out_channels=3, block_out_channels=(8, 16, 16, 32), latent_channels=16,
                    layers_per_block=1, norm_eps=1e-6, norm_num_groups=4, temporal_compression_ratio=4,
                    scaling_factor=0.7, invert_scale_latents=False)


def gen_cog_pipe():
    """a13 pinned to a run of the REFERENCE pipeline's own `__call__`
    (pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py:604-959, unmodified): `prepare_latents` (:350-423), the
    trajectory / identity conditioning (:803-826, through the training script's `img_tensor_to_vae_latent`), the RoPE
    extension (:834-839), the loop with DDIM, DDIM + `use_dynamic_cfg`, and the DPM sampler, decode + post-processing.
    Third-party pieces come from the stand-in (VAE = oracle/cog_vae.py's restatement, schedulers restated: unpinned);
    the VAE's posterior is made (numerically) deterministic -- logvar = -30 -- so that `.sample()` on the global RNG
    (:812, train_code :536) does not enter the fixture; `add_ID_reference_augment_noise` is recorded OFF."""
    import PIL.Image
    import accelerate  # noqa: F401  (installed here; imported before the placeholders exist so that its own optional-
    import transformers  # noqa: F401   dependency probes -- torchvision, ... -- see the truth)
    from transformers import AutoTokenizer, T5EncoderModel, T5Tokenizer  # noqa: F401
    from diffusers.models import AutoencoderKLCogVideoX
    from diffusers.schedulers import CogVideoXDDIMScheduler, CogVideoXDPMScheduler
    from architecture.cogvideox_transformer_3d import CogVideoXTransformer3DModel
    from pipelines.pipeline_cogvideox_i2v_motion_FrameINO import CogVideoXImageToVideoPipeline
    sys.meta_path.append(_PlaceholderFinder())                  # from here on: the training script's imports (:816)
    sys.path.insert(0, REPO)
    from oracle import cog_vae as V

    torch.manual_seed(0)
    dit = CogVideoXTransformer3DModel(**COG_PIPE_DIT).eval()
    randomize_(dit, 61, std=0.12)
    with torch.no_grad():
        dit.patch_embed.pos_embedding.copy_(torch.randn(dit.patch_embed.pos_embedding.shape,
                                                        generator=torch.Generator().manual_seed(62)) * 0.3)
    vae_sd = V.cog_vae_random_state_dict(COG_PIPE_VAE, seed=63)
    lc = COG_PIPE_VAE["latent_channels"]
    vae_sd["encoder.conv_out.conv.weight"][lc:] = 0.0            # logvar rows: posterior std = exp(-15)
    vae_sd["encoder.conv_out.conv.bias"][lc:] = -30.0
    vae = AutoencoderKLCogVideoX(**COG_PIPE_VAE).eval()
    vae.load_flat_state_dict(vae_sd)

    H = W = 64
    F = 9
    g = torch.Generator().manual_seed(65)
    img = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy()
    traj = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    idt = torch.rand(3, H, W, generator=g) * 2 - 1
    pe, ne = torch.randn(1, 8, 16, generator=g), torch.randn(1, 8, 16, generator=g)
    lat0 = torch.randn(1, 3, lc, H // 8, W // 8, generator=g)
    steps, gs = 4, 6.0

    def run(sched, seen=None, dit=dit, vae=vae, pe=pe, ne=ne, lat0=lat0, text=None, **kw):
        pipe = CogVideoXImageToVideoPipeline(tokenizer=text and text[0], text_encoder=text and text[1], vae=vae,
                                             transformer=dit, scheduler=sched)
        if text is not None:             # the prompt= route (:757 -> encode_prompt :269-348 -> _get_t5_prompt_embeds :226-267)
            seen["pe_text"], seen["ne_text"] = pipe.encode_prompt(COG_PROMPT, "", True, 1, max_sequence_length=8)
            return pipe(image=PIL.Image.fromarray(img), traj_tensor=traj, ID_tensor=idt, prompt=COG_PROMPT,
                        negative_prompt="", max_sequence_length=8, height=H, width=W, num_frames=F,
                        num_inference_steps=steps, guidance_scale=gs, add_ID_reference_augment_noise=False,
                        latents=lat0.clone(), **kw).frames
        if seen is not None:
            orig_prep, orig_fwd = pipe.prepare_latents, dit.forward

            def spy_prep(*a, **k):
                out = orig_prep(*a, **k)
                seen["latents_scaled"], seen["image_latents"] = out
                return out

            def spy_fwd(*a, **k):
                if "model_input0" not in seen:
                    seen["model_input0"] = k["hidden_states"].clone()
                    seen["rope_cos"], seen["rope_sin"] = k["image_rotary_emb"]
                return orig_fwd(*a, **k)

            pipe.prepare_latents, dit.forward = spy_prep, spy_fwd
        torch.manual_seed(7)                                       # the global RNG the un-seeded .sample() calls draw from
        try:
            return pipe(image=PIL.Image.fromarray(img), traj_tensor=traj, ID_tensor=idt, prompt_embeds=pe,
                        negative_prompt_embeds=ne, height=H, width=W, num_frames=F, num_inference_steps=steps,
                        guidance_scale=gs, add_ID_reference_augment_noise=False, latents=lat0.clone(), **kw).frames
        finally:
            if seen is not None:
                dit.forward = orig_fwd

    seen = {}
    out_ddim = run(CogVideoXDDIMScheduler(), seen, output_type="latent")
    out_dyn = run(CogVideoXDDIMScheduler(), output_type="latent", use_dynamic_cfg=True)
    out_dpm = run(CogVideoXDPMScheduler(), output_type="latent", generator=torch.Generator().manual_seed(11))
    out_dpm_dyn = run(CogVideoXDPMScheduler(), output_type="latent", use_dynamic_cfg=True,
                      generator=torch.Generator().manual_seed(11))
    video = run(CogVideoXDDIMScheduler(), output_type="np")
    # the call both canonical callers make: prompt strings (app.py:719, test_code/run_cogvideox_FrameIn_mass_evaluation.py:
    # 206-213) through a toy T5EncoderModel (the real transformers class) and a character tokenizer (tests/text_stub.py)
    from tests.text_stub import CharTokenizer, tiny_text_encoder
    te = tiny_text_encoder("t5", seed=72)
    seen_t = {}
    torch.manual_seed(7)
    out_prompt = run(CogVideoXDDIMScheduler(), seen_t, text=(CharTokenizer(), te), output_type="latent")
    # the same calls with every module and the prompt embeddings in bf16 (this pipeline hands the VAE images in the prompt
    # embeddings' dtype, :786-788, so its modules share one dtype -- test_code/run_cogvideox_FrameIn_mass_evaluation.py
    # loads them all in fp16): the reference's own reduced-precision arithmetic and, for DPM, its noise drawn in bf16
    import copy
    dit_b = copy.deepcopy(dit).to(torch.bfloat16)
    vae_b = AutoencoderKLCogVideoX(**COG_PIPE_VAE).eval()
    vae_b.load_flat_state_dict({k: v.bfloat16() for k, v in vae_sd.items()})
    kwb = dict(dit=dit_b, vae=vae_b, pe=pe.bfloat16(), ne=ne.bfloat16(), lat0=lat0.bfloat16(), output_type="latent")
    out_ddim_b = run(CogVideoXDDIMScheduler(), **kwb)
    out_dpm_b = run(CogVideoXDPMScheduler(), generator=torch.Generator().manual_seed(11), **kwb)
    # the reference's OWN bf16 video: what a reduced-precision run of the same pipeline scores against its fp32 video is
    # the yardstick for the HIP pipeline's video
    video_b = run(CogVideoXDDIMScheduler(), **dict(kwb, output_type="np"))
    # round 6: all-fp16 -- transformer, VAE and prompt embeddings in fp16 exactly as the evaluation script loads them
    # (test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-94,106).  Outputs only -> cog_pipe_fp16_tiny.npz.
    dit_h = copy.deepcopy(dit).to(torch.float16)
    vae_h = AutoencoderKLCogVideoX(**COG_PIPE_VAE).eval()
    vae_h.load_flat_state_dict({k: v.half() for k, v in vae_sd.items()})
    kwh = dict(dit=dit_h, vae=vae_h, pe=pe.half(), ne=ne.half(), lat0=lat0.half(), output_type="latent")
    out_ddim_h = run(CogVideoXDDIMScheduler(), **kwh)
    out_dyn_h = run(CogVideoXDDIMScheduler(), use_dynamic_cfg=True, **kwh)
    out_dpm_h = run(CogVideoXDPMScheduler(), generator=torch.Generator().manual_seed(11), **kwh)
    video_h = run(CogVideoXDDIMScheduler(), **dict(kwh, output_type="np"))
    save("cog_pipe_fp16_tiny", out_ddim_fp16=out_ddim_h, out_ddim_dynamic_cfg_fp16=out_dyn_h, out_dpm_fp16=out_dpm_h,
         out_video_fp16=video_h)
    x0 = seen["model_input0"]                                       # [2, 4, 48, 8, 8] = [noisy + ID | first frame + 0 | traj + 0]
    print("placeholder modules served:", sorted(set(_PlaceholderFinder.served)))
    sd = {"dit." + k: v for k, v in dit.state_dict().items()}
    sd.update({"vae." + k: v for k, v in vae_sd.items()})
    cfg = dict(COG_PIPE_DIT)
    cfg.update({"vae_" + k: v for k, v in COG_PIPE_VAE.items()})
    save("cog_pipe_tiny", cfg=cfg, sd=sd, image=img, traj=traj, id_tensor=idt, prompt_embeds=pe, negative_embeds=ne,
         latents0=lat0, latents_scaled=seen["latents_scaled"], image_latents=seen["image_latents"],
         traj_latents=x0[1:2, :3, 2 * lc:], id_latent=x0[1:2, 3:, :lc], model_input0=x0, rope_cos=seen["rope_cos"],
         rope_sin=seen["rope_sin"], out_ddim=out_ddim, out_ddim_dynamic_cfg=out_dyn, out_dpm=out_dpm,
         out_dpm_dynamic_cfg=out_dpm_dyn, out_ddim_bf16=out_ddim_b, out_dpm_bf16=out_dpm_b, out_video=video,
         out_video_bf16=video_b, steps=np.array(steps), guidance=np.array(gs), dpm_generator_seed=np.array(11),
         prompt=np.array(COG_PROMPT), negative_prompt=np.array(""), prompt_embeds_from_text=seen_t["pe_text"],
         negative_embeds_from_text=seen_t["ne_text"], out_ddim_prompt=out_prompt,
         **{"te/" + k: v for k, v in te.state_dict().items()})


def gen_blend():
    """The tile blends of the VAE tiling: the reference's in-tree blend_v / blend_h (architecture/autoencoder_kl_wan.py:1254-1268,
    the same loops diffusers' CogVideoX VAE runs) on random 5-D tiles, extents below / at / above the tile size."""
    from architecture.autoencoder_kl_wan import AutoencoderKLWan
    g = torch.Generator().manual_seed(31)
    arrays = {}
    for i, (shape, extent) in enumerate([((1, 3, 2, 6, 5), 2), ((1, 2, 3, 8, 7), 5), ((2, 4, 1, 5, 9), 9), ((1, 1, 2, 4, 4), 1)]):
        a = torch.randn(*shape, generator=g)
        b = torch.randn(*shape, generator=g)
        arrays[f"a_{i}"], arrays[f"b_{i}"], arrays[f"extent_{i}"] = a, b, np.array(extent)
        arrays[f"v_{i}"] = AutoencoderKLWan.blend_v(None, a.clone(), b.clone(), extent)
        arrays[f"h_{i}"] = AutoencoderKLWan.blend_h(None, a.clone(), b.clone(), extent)
    save("vae_blend", **arrays)


GENS = {"blend": gen_blend, "wan_dit": gen_wan_dit, "wan_pipe": gen_wan_pipe, "wan_vae": gen_wan_vae, "cog_dit": gen_cog_dit,
        "cog_dit_s1": gen_cog_dit_s1, "cog_pipe": gen_cog_pipe, "traj_kernel": gen_traj_kernel}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for k, fn in GENS.items():
        if not a.only or k in a.only.split(","):
            fn()
