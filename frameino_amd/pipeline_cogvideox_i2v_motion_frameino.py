"""MI355X-native FrameINO CogVideoX image-to-video pipeline (mirror of the reference's
pipelines/pipeline_cogvideox_i2v_motion_FrameINO.py::CogVideoXImageToVideoPipeline, denoising side).

What is built: the CFG-batched denoise loop (:848-944) -- model-input assembly
`[noisy + ID | first-frame + 0 | trajectory + 0]` on the channel axis (:856-881), one B=2 DiT forward per step on HIP
kernels, ID-frame drop (:901-902), (dynamic) guidance (:906-911), v-prediction DDIM update (:916) in one kernel --
plus `_prepare_rotary_positional_embeddings` (:540-584) with the FrameIn extension (:834-839).

`__call__` (:604-957) takes what the reference's takes: `prompt=` / `negative_prompt=` strings go through `encode_prompt`
(:269-348) and `_get_t5_prompt_embeds` (:226-267) on the pipeline's `tokenizer` / `text_encoder` -- the HF `transformers`
T5 objects, third-party on both sides, once per clip -- or pre-computed `prompt_embeds`; the condition encodes and the
decode run on the `vae` handed in (frameino_amd.autoencoder_kl_cogvideox, or any object with the diffusers
AutoencoderKLCogVideoX interface); `denoise()` works on latents alone.
"""
import math
from types import SimpleNamespace

import torch

from . import ops
from .schedulers import CogVideoXDDIMScheduler  # noqa: F401  (re-export)


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """reference :72-89"""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        resize_height = th
        resize_width = int(round(th / h * w))
    else:
        resize_width = tw
        resize_height = int(round(tw / w * h))
    crop_top = int(round((th - resize_height) / 2.0))
    crop_left = int(round((tw - resize_width) / 2.0))
    return (crop_top, crop_left), (crop_top + resize_height, crop_left + resize_width)


def _rope_1d(dim, pos, theta=10000.0):
    """architecture/embeddings.py:1153-1216 (use_real, repeat_interleave_real, fp32 freqs)"""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos, freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta=10000):
    """architecture/embeddings.py:864-962, grid_type="linspace" (CogVideoX 1.0)."""
    start, stop = crops_coords
    gh, gw = grid_size
    grid_h = torch.linspace(start[0], stop[0] * (gh - 1) / gh, gh, dtype=torch.float32)
    grid_w = torch.linspace(start[1], stop[1] * (gw - 1) / gw, gw, dtype=torch.float32)
    grid_t = torch.linspace(0, temporal_size * (temporal_size - 1) / temporal_size, temporal_size, dtype=torch.float32)
    dim_t, dim_h, dim_w = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3
    ft, fh, fw = _rope_1d(dim_t, grid_t, theta), _rope_1d(dim_h, grid_h, theta), _rope_1d(dim_w, grid_w, theta)

    def combine(a, b, c):
        a = a[:, None, None, :].expand(-1, gh, gw, -1)
        b = b[None, :, None, :].expand(temporal_size, -1, gw, -1)
        c = c[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([a, b, c], dim=-1).reshape(temporal_size * gh * gw, -1)

    return combine(ft[0], fh[0], fw[0]), combine(ft[1], fh[1], fw[1])


class CogVideoXPipelineOutput(SimpleNamespace):
    pass


def _one_generator(generator, batch_size=1):
    """diffusers' randn_tensor rule for a list of generators: its length must equal the batch size, and a list of one
    IS that generator -- never silently dropped (batches: the callers hand every video its own generator)."""
    if isinstance(generator, (list, tuple)):
        if len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length of "
                             f"the generators.")
        if batch_size != 1:
            raise ValueError("a list of generators is consumed one per video: index it before drawing a single row")
        return generator[0]
    return generator


class CogVideoXImageToVideoPipeline:
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]
    extend_rope_by_first_frame = True             # :834-839 (the ID frame reuses the first frame's RoPE rows)

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, transformer=None, scheduler=None):
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.transformer, self.scheduler = transformer, scheduler
        self.vae_scale_factor_spatial = 8
        self.vae_scale_factor_temporal = 4
        self.vae_scaling_factor_image = getattr(getattr(vae, "config", None), "scaling_factor", 0.7)
        self._interrupt = False
        self.use_hip_graph = None            # None: replay the step from a captured hipGraph whenever the loop is
        #                                      capturable (no per-step callback); False: eager; True: capture or raise

    def enable_model_cpu_offload(self, *a, **k):
        return self

    def to(self, device):
        for m in (self.transformer, self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, **kwargs):
        """`CogVideoXImageToVideoPipeline.from_pretrained(base_folder, text_encoder=, transformer=, vae=, torch_dtype=)`
        (reference test_code/run_cogvideox_FrameIn_mass_evaluation.py:101-107): components handed in are used as they are,
        the others come from the sub-folders of a LOCAL copy of zai-org/CogVideoX-5b-I2V that exist."""
        from . import loading
        from .autoencoder_kl_cogvideox import AutoencoderKLCogVideoX
        from .cogvideox_transformer_3d import CogVideoXTransformer3DModel
        comps = {"transformer": lambda f, dt: CogVideoXTransformer3DModel.from_pretrained(f, torch_dtype=dt),
                 "vae": lambda f, dt: AutoencoderKLCogVideoX.from_pretrained(f, torch_dtype=dt),
                 "scheduler": lambda f, dt: loading.load_scheduler(f),
                 "text_encoder": loading._load_text_encoder, "tokenizer": loading._load_tokenizer}
        parts, index, rest = loading._pipeline_from_pretrained(cls, pretrained_model_name_or_path, comps, torch_dtype,
                                                               **kwargs)
        if rest:
            raise TypeError(f"{cls.__name__}.from_pretrained: unexpected keyword arguments {sorted(rest)}")
        return cls(tokenizer=parts["tokenizer"], text_encoder=parts["text_encoder"], vae=parts["vae"],
                   transformer=parts["transformer"], scheduler=parts["scheduler"])

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def _execution_device(self):
        return self.transformer.device

    # ---- text (once per clip; HF transformers T5, not on the kernel path) ----
    def _get_t5_prompt_embeds(self, prompt=None, num_videos_per_prompt=1, max_sequence_length=226, device=None,
                              dtype=None):
        """reference :226-267: tokenizer(padding="max_length", truncation) -> text_encoder(ids)[0]; unlike the Wan
        pipeline no attention mask is passed and the padded positions keep what the encoder made of them"""
        if self.tokenizer is None or self.text_encoder is None:
            raise ValueError("encoding a prompt string needs the pipeline's `tokenizer` and `text_encoder` (the HF T5 "
                             "objects the reference loads, test_code/run_cogvideox_FrameIn_mass_evaluation.py:92-101); "
                             "pass them, or pass `prompt_embeds` / `negative_prompt_embeds`")
        device = device or self._execution_device
        dtype = dtype or self.text_encoder.dtype
        prompt = [prompt] if isinstance(prompt, str) else prompt
        bsz = len(prompt)
        ti = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                            add_special_tokens=True, return_tensors="pt")
        ids = ti.input_ids
        untruncated = self.tokenizer(prompt, padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] >= ids.shape[-1] and not torch.equal(ids, untruncated):
            import warnings
            removed = self.tokenizer.batch_decode(untruncated[:, max_sequence_length - 1:-1])
            warnings.warn(f"The following part of your input was truncated because `max_sequence_length` is set to "
                          f"{max_sequence_length} tokens: {removed}")
        emb = self.text_encoder(ids.to(device))[0].to(dtype=dtype, device=device)
        _, seq_len, _ = emb.shape
        return emb.repeat(1, num_videos_per_prompt, 1).view(bsz * num_videos_per_prompt, seq_len, -1)

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance=True, num_videos_per_prompt=1,
                      prompt_embeds=None, negative_prompt_embeds=None, max_sequence_length=226, device=None, dtype=None):
        """reference :269-348"""
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        bsz = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self._get_t5_prompt_embeds(prompt, num_videos_per_prompt, max_sequence_length, device, dtype)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt = bsz * [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got "
                                f"{type(negative_prompt)} != {type(prompt)}.")
            elif bsz != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but "
                                 f"`prompt`: {prompt} has batch size {bsz}. Please make sure that passed "
                                 f"`negative_prompt` matches the batch size of `prompt`.")
            negative_prompt_embeds = self._get_t5_prompt_embeds(negative_prompt, num_videos_per_prompt,
                                                                max_sequence_length, device, dtype)
        return prompt_embeds, negative_prompt_embeds

    def check_inputs(self, image, prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs,
                     latents=None, prompt_embeds=None, negative_prompt_embeds=None):
        """reference :461-523 (same conditions, same error classes)"""
        import PIL.Image
        if not isinstance(image, (torch.Tensor, PIL.Image.Image, list)):
            raise ValueError(f"`image` has to be of type `torch.Tensor` or `PIL.Image.Image` or `List[PIL.Image.Image]` "
                             f"but is {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but "
                             f"found {bad}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of "
                             "the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` "
                             "undefined.")
        elif prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `negative_prompt_embeds`. Please make sure to only forward "
                             "one of the two.")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`. Please make sure to "
                             "only forward one of the two.")
        if prompt_embeds is not None and negative_prompt_embeds is not None \
                and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError(f"`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed "
                             f"directly, but got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` "
                             f"{negative_prompt_embeds.shape}.")

    def _prepare_rotary_positional_embeddings(self, height, width, num_frames, device):
        """reference :540-584 (patch_size_t None) + the FrameIn first-frame extension :834-839."""
        c = self.transformer.config
        gh = height // (self.vae_scale_factor_spatial * c.patch_size)
        gw = width // (self.vae_scale_factor_spatial * c.patch_size)
        crops = get_resize_crop_region_for_grid((gh, gw), c.sample_width // c.patch_size, c.sample_height // c.patch_size)
        cos, sin = get_3d_rotary_pos_embed(c.attention_head_dim, crops, (gh, gw), num_frames)
        if self.extend_rope_by_first_frame:
            n1 = cos.shape[0] // num_frames
            cos = torch.cat([cos, cos[:n1]], dim=0)
            sin = torch.cat([sin, sin[:n1]], dim=0)
        return cos.to(device), sin.to(device)

    @torch.no_grad()
    def denoise(self, latents, image_latents, traj_latents, id_latent, prompt_embeds, negative_prompt_embeds,
                guidance_scale=6.0, num_inference_steps=50, use_dynamic_cfg=False, image_rotary_emb=None,
                attention_kwargs=None, callback_on_step_end=None, generator=None):
        """reference :848-944.  latents / image_latents / traj_latents [1, F, C, h, w], id_latent [1, 1, C, h, w] or
        None; returns the final latents [1, F, C, h, w] in the transformer dtype."""
        tr = self.transformer
        dev, dt = latents.device, tr.dtype
        if latents.shape[0] != 1:
            # A batch (a list of prompts, batched `prompt_embeds` / `latents`) runs video by video: the loop state -- the static
            # model-input buffer, the sampler history, the captured graph -- is one video's.  The videos of a batched loop never
            # meet (per-sample attention, per-sample guidance), so every row equals the single call on that row's noise, prompt
            # and conditions; conditions given with batch 1 (the reference's traj / ID latents are batch 1, :803-826) are shared.
            nb_ = latents.shape[0]

            def row(t, i):
                return t if t is None or t.shape[0] == 1 else t[i:i + 1]
            outs = []
            for i in range(nb_):
                gi = generator[i] if isinstance(generator, (list, tuple)) else generator
                outs.append(self.denoise(latents[i:i + 1], row(image_latents, i), row(traj_latents, i), row(id_latent, i),
                                         row(prompt_embeds, i), row(negative_prompt_embeds, i), guidance_scale,
                                         num_inference_steps, use_dynamic_cfg, image_rotary_emb, attention_kwargs,
                                         callback_on_step_end, gi))
            return torch.cat(outs, dim=0)
        generator = _one_generator(generator)
        cfg_on = guidance_scale > 1.0 and negative_prompt_embeds is not None
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        ts = self.scheduler.timesteps
        nlf = latents.shape[1]
        lat = (latents * self.scheduler.init_noise_sigma).to(dt)[0].contiguous().clone()      # [F, C, h, w]
        img, trj = image_latents.to(dt), traj_latents.to(dt)
        if id_latent is not None:
            idl = id_latent.to(dt)
            pad = torch.zeros_like(idl)
            img, trj = torch.cat([img, pad], dim=1), torch.cat([trj, pad], dim=1)              # :874-876
        nb = 2 if cfg_on else 1
        prompt = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(dt) if cfg_on else prompt_embeds.to(dt)
        if image_rotary_emb is None:
            h = latents.shape[3] * self.vae_scale_factor_spatial
            w = latents.shape[4] * self.vae_scale_factor_spatial
            image_rotary_emb = self._prepare_rotary_positional_embeddings(h, w, nlf, dev)
        # host-side schedule of the guidance weight (:906-909; the reference calls t.item() per step)
        n = num_inference_steps
        gs = [1 + guidance_scale * ((1 - math.cos(math.pi * ((n - int(t)) / n) ** 5.0)) / 2) if use_dynamic_cfg
              else guidance_scale for t in ts.tolist()]
        gcol = torch.tensor(gs, dtype=torch.float32, device=dev)[:, None]
        dpm = getattr(self.scheduler, "kind", "ddim") == "dpm"          # CogVideoXDPMScheduler branch (:915-926)
        sc = self.scheduler.coefs.to(dev)
        coefs = torch.cat([sc[:, :7], gcol, sc[:, 7:8]], 1) if dpm else torch.cat([sc, gcol], 1)

        # Static buffers of the loop (what a captured step reads and writes).  The model input [nb, F(+1), 3C, h, w] =
        # [noisy | first frame + 0 | trajectory + 0] on the channel axis, identity frame appended on the frame axis
        # (:866-880): only the noisy C channels of the generated frames change from step to step, so everything else is
        # written once here and each step copies `lat` into its slot (the reference re-concatenates all of it).
        C = lat.shape[1]
        nf_in = nlf + (0 if id_latent is None else id_latent.shape[1])
        st = SimpleNamespace(lat=lat, nlf=nlf, C=C, cfg_on=cfg_on, dpm=dpm, prompt=prompt, rot=image_rotary_emb,
                             attention_kwargs=attention_kwargs)
        st.x = torch.zeros((nb, nf_in, 3 * C) + tuple(lat.shape[2:]), dtype=dt, device=dev)
        if id_latent is not None:
            st.x[:, nlf:, :C] = idl                                                             # :868 (img / trj pads stay 0)
        st.x[:, :, C:2 * C] = img                              # (already zero-padded on the ID frame, :874-876)
        st.x[:, :, 2 * C:] = trj
        st.t = torch.zeros(nb, dtype=ts.dtype, device=dev)
        st.coef = torch.zeros(coefs.shape[1], dtype=torch.float32, device=dev)
        st.x0_old = torch.zeros(lat.shape, dtype=torch.float32, device=dev) if dpm else None
        st.noise = torch.zeros_like(lat) if dpm else None
        from .graph_step import StepGraph
        from .pipeline_wan_i2v_motion_frameino import tr_default_procs
        stepper = StepGraph(lambda: self._step(st), self.use_hip_graph,
                            callback_on_step_end is None and lat.is_cuda
                            and (self.use_hip_graph is True or tr_default_procs(tr)), len(ts))
        for i, t in enumerate(ts):
            if self._interrupt:
                continue
            st.t.copy_(t.expand(nb))                          # device-to-device: no host sync
            st.coef.copy_(coefs[i])
            if dpm:
                st.noise.copy_(self.scheduler.noise(i, lat.shape, generator, dev, dt))
            stepper.step()
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": lat[None]})
                if "latents" in out and out["latents"] is not None:
                    lat.copy_(out["latents"][0])
        stepper.close()
        return lat[None]

    def _step(self, st):
        """One denoise step on the static buffers `st` (no host sync, no data-dependent shapes: hipGraph-capturable)."""
        st.x[:, :st.nlf, :st.C].copy_(st.lat)                                                   # broadcast over the CFG batch
        # (live_frames: the identity frame appended on the frame axis is dropped from the prediction, :896 -- the model may skip
        # what only that frame's output needs; only passed to the mirror's own transformer class)
        live = {"live_frames": st.nlf} if (st.x.shape[1] > st.nlf and getattr(self.transformer, "skip_dead_rows", False)) else {}
        pred = self.transformer(hidden_states=st.x, encoder_hidden_states=st.prompt, timestep=st.t,
                                image_rotary_emb=st.rot, attention_kwargs=st.attention_kwargs, return_dict=False, **live)[0]
        if st.dpm:
            ops.cfg_dpm_step_(st.lat, pred.contiguous(), st.x0_old, st.noise, st.coef, has_uncond=st.cfg_on)
        else:
            ops.cfg_vpred_step_(st.lat, pred.contiguous(), st.coef, has_uncond=st.cfg_on)       # :896-927

    # ---- condition encodes / decode around a user-supplied VAE (diffusers interface) ----
    def _need_vae(self):
        if self.vae is None:
            raise NotImplementedError(
                "CogVideoXImageToVideoPipeline.__call__ needs a `vae` with the diffusers AutoencoderKLCogVideoX "
                "interface (.encode(x).latent_dist.sample(generator), .decode(z).sample, .config.scaling_factor, "
                ".config.invert_scale_latents, .dtype): that model is third-party and not part of the reference tree. "
                "`denoise()` works on latents alone.")

    @staticmethod
    def _noise_rows(shape, generator, dtype, device):
        """diffusers' randn_tensor: one generator draws the whole batch in one call, a list of generators one row each (its
        length must be the batch size); drawn on the generator's device, then moved"""
        if isinstance(generator, (list, tuple)) and shape[0] > 1:
            if len(generator) != shape[0]:
                raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                                 f"effective batch size of {shape[0]}. Make sure the batch size matches the length of "
                                 f"the generators.")
            return torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=g_, device=g_.device, dtype=dtype).to(device)
                              for g_ in generator], dim=0)
        g1 = _one_generator(generator, shape[0])
        gdev = g1.device if g1 is not None else device
        return torch.randn(tuple(shape), generator=g1, device=gdev, dtype=dtype).to(device)

    def prepare_latents(self, image, batch_size=1, num_channels_latents=16, num_frames=13, height=60, width=90,
                        dtype=None, device=None, generator=None, latents=None):
        """reference :350-423 (patch_size_t None).  image [B, C, H, W] in [-1, 1]."""
        self._need_vae()
        if isinstance(generator, tuple):
            generator = list(generator)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length of "
                             f"the generators.")
        nlf = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        lh, lw = height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial
        image = image.unsqueeze(2)                                                   # [B, C, 1, H, W]
        gens = generator if isinstance(generator, list) else [generator] * image.shape[0]
        image_latents = torch.cat([self.vae.encode(img.unsqueeze(0)).latent_dist.sample(g)
                                   for img, g in zip(image, gens)], dim=0).to(dtype).permute(0, 2, 1, 3, 4)
        if not getattr(self.vae.config, "invert_scale_latents", False):
            image_latents = self.vae_scaling_factor_image * image_latents
        else:
            image_latents = 1 / self.vae_scaling_factor_image * image_latents
        pad = torch.zeros((batch_size, nlf - 1, num_channels_latents, lh, lw), device=device, dtype=dtype)
        image_latents = torch.cat([image_latents, pad], dim=1)
        if latents is None:
            latents = self._noise_rows((batch_size, nlf, num_channels_latents, lh, lw), generator, dtype, device)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma, image_latents

    def decode_latents(self, latents):
        """reference :426-431"""
        self._need_vae()
        latents = latents.permute(0, 2, 1, 3, 4)
        return self.vae.decode(1 / self.vae_scaling_factor_image * latents).sample

    @torch.no_grad()
    def __call__(self, image, traj_tensor=None, ID_tensor=None, prompt=None, negative_prompt=None, height=None,
                 width=None, num_frames=49, num_inference_steps=50, timesteps=None, guidance_scale=6,
                 use_dynamic_cfg=False, add_ID_reference_augment_noise=True, num_videos_per_prompt=1, eta=0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type="pil",
                 return_dict=True, attention_kwargs=None, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=["latents"], max_sequence_length=226):
        """reference :604-957: `prompt=` / `negative_prompt=` strings (what app.py:719 and
        test_code/run_cogvideox_FrameIn_mass_evaluation.py:206-213 pass) or pre-computed `prompt_embeds` /
        `negative_prompt_embeds` [B, 226, 4096]; a list of B prompts / B embedding rows gives B videos (run one by one; the
        reference's own loop only survives B = 1).  `num_videos_per_prompt` is accepted and, as in the reference (:723), ignored.  As in the reference (:744, :766-768) guidance needs a negative branch:
        with `guidance_scale > 1` and no `negative_prompt_embeds` the empty negative prompt is encoded."""
        self._need_vae()
        if timesteps is not None or eta != 0.0:
            raise NotImplementedError("custom timesteps / eta: the built sampler is the v-prediction DDIM step (eta 0)")
        if hasattr(callback_on_step_end, "tensor_inputs"):
            callback_on_step_end_tensor_inputs = callback_on_step_end.tensor_inputs
        c = self.transformer.config
        height = height or c.sample_height * self.vae_scale_factor_spatial
        width = width or c.sample_width * self.vae_scale_factor_spatial
        self.check_inputs(image, prompt, height, width, negative_prompt, callback_on_step_end_tensor_inputs, latents,
                          prompt_embeds, negative_prompt_embeds)
        # The reference overwrites `num_videos_per_prompt` with 1 (:723) and its loop breaks on more than one prompt (the
        # trajectory / identity latents are batch 1, :803-826 against torch.cat at :872-880); here a list of B prompts (or
        # B rows of `prompt_embeds`) makes B videos, run video by video in `denoise` (the Wan mirror does the same).
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        dev = self._execution_device
        dt = self.transformer.dtype
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, negative_prompt, guidance_scale > 1.0, 1, prompt_embeds, negative_prompt_embeds, max_sequence_length,
            dev)                                                                                # :757-766
        prompt_embeds = prompt_embeds.to(dev)
        self._interrupt = False
        if isinstance(image, torch.Tensor):
            img = image if image.ndim == 4 else image[None]
            img = img if img.min() < 0 else 2.0 * img - 1.0
        else:
            import numpy as np
            import PIL.Image
            pil = image if isinstance(image, list) else [image]
            arrs = [np.asarray(im.resize((width, height), resample=PIL.Image.LANCZOS)).astype("float32") / 255.0 for im in pil]
            img = 2.0 * torch.from_numpy(np.stack(arrs).transpose(0, 3, 1, 2).copy()) - 1.0
        img = img.to(dev, dtype=dt)
        if img.shape[0] not in (1, batch_size):
            raise ValueError(f"`image` holds {img.shape[0]} images for {batch_size} prompts: pass one image, or one per prompt")
        n_lat = c.out_channels          # 16 for CogVideoX-5B (hard-coded at :797)
        if img.shape[0] == 1 and batch_size > 1:
            # ONE first frame for B prompts: its latent is one posterior sample, shared by the B videos like the trajectory and
            # identity latents below; the noise is drawn as the batch's rows (diffusers' randn_tensor rule for generator lists).
            # With a LIST of generators only row 0 therefore reproduces the single call made with its generator (posterior draw
            # first, then the noise row): rows i > 0 draw their noise from generator[i] with no posterior draw in front of it.
            g0 = generator[0] if isinstance(generator, (list, tuple)) else generator
            _, image_latents = self.prepare_latents(img, 1, n_lat, num_frames, height, width, dt, dev, g0,
                                                    torch.zeros(1, device=dev))
            nlf_ = (num_frames - 1) // self.vae_scale_factor_temporal + 1
            shape = (batch_size, nlf_, n_lat, height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
            latents = self._noise_rows(shape, generator, dt, dev) if latents is None else latents.to(dev)
            latents = latents * self.scheduler.init_noise_sigma
        else:
            latents, image_latents = self.prepare_latents(img, batch_size, n_lat, num_frames, height, width, dt, dev, generator,
                                                          latents)
        latents = latents / self.scheduler.init_noise_sigma          # denoise() applies it (:421)
        vdt = getattr(self.vae, "dtype", dt)
        traj = traj_tensor.to(dev, dtype=vdt)[None].permute(0, 2, 1, 3, 4)                        # :809-811
        traj_latents = self.vae.encode(traj).latent_dist.sample() * self.vae.config.scaling_factor
        traj_latents = traj_latents.permute(0, 2, 1, 3, 4).contiguous().float().to(dt)            # [1, F, C, h, w]
        id_latent = None
        if ID_tensor is not None:                                                                  # :820-826, train_code :515-546
            idt = ID_tensor.unsqueeze(0).unsqueeze(2).to(dev, dtype=vdt)                           # [1, C, 1, H, W]
            if add_ID_reference_augment_noise:
                sigma = torch.exp(torch.normal(mean=-3.0, std=0.5, size=(1,), device=dev)).to(idt.dtype)
                idt = idt + torch.randn_like(idt) * sigma[:, None, None, None, None]
            idl = self.vae.encode(idt).latent_dist.sample() * self.vae.config.scaling_factor
            id_latent = idl.squeeze(2).contiguous().float().unsqueeze(1).to(dt)                    # [1, 1, C, h, w]
        gscale = guidance_scale if negative_prompt_embeds is not None else 1.0
        out = self.denoise(latents, image_latents, traj_latents, id_latent, prompt_embeds, negative_prompt_embeds,
                           gscale, num_inference_steps, use_dynamic_cfg, None, attention_kwargs, callback_on_step_end,
                           generator)
        if output_type == "latent":
            video = out
        else:
            frames = torch.cat([self.decode_latents(out[i:i + 1]) for i in range(out.shape[0])])   # [B, C, F, H, W]
            v = (frames.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)                      # [B, F, C, H, W]
            if output_type == "pt":
                video = v
            elif output_type == "np":
                video = v.permute(0, 1, 3, 4, 2).cpu().numpy()
            elif output_type == "pil":
                import PIL.Image
                arr = (v.permute(0, 1, 3, 4, 2).cpu().numpy() * 255).round().astype("uint8")
                video = [[PIL.Image.fromarray(f) for f in vid] for vid in arr]
            else:
                raise ValueError(f"unsupported output_type {output_type}")
        if not return_dict:
            return (video,)
        return CogVideoXPipelineOutput(frames=video)
