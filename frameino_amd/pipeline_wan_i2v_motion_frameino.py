"""MI355X-native FrameINO Wan2.2 image-to-video pipeline (drop-in for the reference's
pipelines/pipeline_wan_i2v_motion_FrameINO.py::WanImageToVideoPipeline on the Wan2.2-TI2V-5B /
`expand_timesteps=True` path, which is the path FrameINO runs).

Same constructor arguments, same `__call__` keyword set (reference :581-609), same `prepare_latents` contract
(:400-553), `.frames` output, `callback_on_step_end`, `interrupt`, `enable_model_cpu_offload()` (a no-op: 288 GB HBM).
The 50-step loop (:809-908) runs on HIP kernels only:

    fino_wan_model_input      mask-blend with the clean first-frame latent, ID frame(s) appended on the frame axis,
                              trajectory latents concatenated on the channel axis (:829, :854, :858)
    WanTransformer3DModel x2  cond / uncond forward with the de-duplicated per-token timestep {0, t} (:832-843)
    fino_cfg_euler_step       CFG combine, ID-frame drop, flow-match Euler update (:882-891)

The loop replays ONE captured step from a hipGraph (frameino_amd/graph_step.py: step 0 eager, step 1 captured, the rest
replays; timestep and dt live in device buffers that a device-to-device copy refreshes between replays, no host sync
inside the loop) whenever the call has no per-step callback: on one GPU always, on token shards for the call patterns this
image's runtime captures (graph_step.py lists them).  `use_hip_graph = False` keeps the eager loop, `True` makes a loop
that cannot be captured an error.
"""
import html
import re
from types import SimpleNamespace

import numpy as np
import torch

from . import ops


def _basic_clean(text):
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


def prompt_clean(text):
    """reference :103-117"""
    return re.sub(r"\s+", " ", _basic_clean(text)).strip()


class VideoProcessor:
    """The two diffusers VideoProcessor calls the pipeline makes (:767, :927); host-side pre/post-processing."""

    def __init__(self, vae_scale_factor=16):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height, width):
        import PIL.Image
        if isinstance(image, PIL.Image.Image):
            image = image.resize((width, height), resample=PIL.Image.LANCZOS)
            arr = np.asarray(image).astype(np.float32) / 255.0
            if arr.ndim == 2:
                arr = arr[..., None]
            return 2.0 * torch.from_numpy(arr.transpose(2, 0, 1).copy())[None] - 1.0
        if isinstance(image, torch.Tensor):
            t = image if image.ndim == 4 else image[None]
            return t if t.min() < 0 else 2.0 * t - 1.0
        raise ValueError(f"`image` has to be of type `torch.Tensor` or `PIL.Image.Image` but is {type(image)}")

    def postprocess_video(self, video, output_type="np"):
        v = (video.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)       # [B, F, C, H, W]
        if output_type == "pt":
            return v
        if output_type == "np":
            return v.permute(0, 1, 3, 4, 2).cpu().numpy()
        raise ValueError(f"unsupported output_type {output_type}")


import contextlib as _contextlib

_null_ctx = _contextlib.nullcontext


class WanPipelineOutput(SimpleNamespace):
    pass


def tr_default_procs(transformer):
    """the built-in processors run only this library's kernels: capturable.  A user-installed processor may do anything
    (host syncs included), so its loop stays eager unless `use_hip_graph = True` asks for the capture"""
    f = getattr(transformer, "_default_processors", None)
    return bool(f()) if callable(f) else True


class WanImageToVideoPipeline:
    model_cpu_offload_seq = "text_encoder->image_encoder->transformer->transformer_2->vae"
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, scheduler=None, image_processor=None,
                 image_encoder=None, transformer=None, transformer_2=None, boundary_ratio=None,
                 expand_timesteps=False):
        if transformer_2 is not None or boundary_ratio is not None or image_encoder is not None:
            raise NotImplementedError("two-stage / Wan2.1 image-encoder variants are outside FrameINO's TI2V-5B path")
        if not expand_timesteps:
            raise NotImplementedError("FrameINO runs Wan2.2-TI2V-5B with expand_timesteps=True (reference :826-843)")
        self.tokenizer, self.text_encoder, self.vae = tokenizer, text_encoder, vae
        self.scheduler, self.transformer = scheduler, transformer
        self.config = SimpleNamespace(boundary_ratio=boundary_ratio, expand_timesteps=expand_timesteps)
        self.vae_scale_factor_temporal = vae.config.scale_factor_temporal if vae is not None else 4
        self.vae_scale_factor_spatial = vae.config.scale_factor_spatial if vae is not None else 8
        self.video_processor = VideoProcessor(vae_scale_factor=self.vae_scale_factor_spatial)
        self.use_hip_graph = None        # None: graph replay whenever the loop is capturable (graph_step.StepGraph)
        self.batch_cfg = True            # run cond+uncond as one batch-2 forward when not CFG-parallel
        self.shard_vae_decode = True     # under a multi-rank plan: every rank decodes a slab of the frame (parallel.sharded_vae_decode)
        self.shard_vae_encode = True     # ... and encodes a slab of the trajectory video (parallel.sharded_vae_encode)
        self.cfg_streams = False         # ... or as two B=1 forwards on two concurrent streams (takes precedence)
        self._streams = None
        self._interrupt = False
        self._graph = None

    # ---- diffusers-style conveniences ----
    @property
    def _execution_device(self):
        return self.transformer.device

    def enable_model_cpu_offload(self, *a, **k):      # reference app.py:163 -- unnecessary with 288 GB of HBM
        return self

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, **kwargs):
        """`WanImageToVideoPipeline.from_pretrained(base_folder, transformer=transformer, vae=vae, torch_dtype=...)`
        (reference app.py:161; train_code/train_wan_motion_FrameINO.py builds it the same way): components handed in are
        used as they are, the others are loaded from the sub-folders of a LOCAL copy of Wan-AI/Wan2.2-TI2V-5B-Diffusers
        that exist (transformer/, vae/, scheduler/ by the class its config names, text_encoder/ + tokenizer/ through
        `transformers`); `expand_timesteps` from model_index.json (true for TI2V-5B, the only path FrameINO runs)."""
        from . import loading
        from .autoencoder_kl_wan import AutoencoderKLWan
        from .transformer_wan import WanTransformer3DModel
        comps = {"transformer": lambda f, dt: WanTransformer3DModel.from_pretrained(f, torch_dtype=dt),
                 "vae": lambda f, dt: AutoencoderKLWan.from_pretrained(f, torch_dtype=dt),
                 "scheduler": lambda f, dt: loading.load_scheduler(f),
                 "text_encoder": loading._load_text_encoder, "tokenizer": loading._load_tokenizer}
        parts, index, rest = loading._pipeline_from_pretrained(cls, pretrained_model_name_or_path, comps, torch_dtype,
                                                               **kwargs)
        for k in ("image_encoder", "image_processor", "transformer_2", "boundary_ratio"):
            rest.pop(k, None)
        if rest:
            raise TypeError(f"{cls.__name__}.from_pretrained: unexpected keyword arguments {sorted(rest)}")
        return cls(tokenizer=parts["tokenizer"], text_encoder=parts["text_encoder"], vae=parts["vae"],
                   scheduler=parts["scheduler"], transformer=parts["transformer"],
                   expand_timesteps=bool(index.get("expand_timesteps", True)))

    def maybe_free_model_hooks(self):
        pass

    def to(self, device):
        for m in (self.transformer, self.vae, self.text_encoder):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def current_timestep(self):
        return self._current_timestep

    @property
    def interrupt(self):
        return self._interrupt

    # ---- text (once per clip; HF transformers model, not on the kernel path) ----
    def _get_t5_prompt_embeds(self, prompt, num_videos_per_prompt=1, max_sequence_length=512, device=None, dtype=None):
        """reference :206-245"""
        device = device or self._execution_device
        dtype = dtype or self.text_encoder.dtype
        prompt = [prompt] if isinstance(prompt, str) else prompt
        prompt = [prompt_clean(u) for u in prompt]
        bsz = len(prompt)
        ti = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                            add_special_tokens=True, return_attention_mask=True, return_tensors="pt")
        ids, mask = ti.input_ids, ti.attention_mask
        seq_lens = mask.gt(0).sum(dim=1).long()
        emb = self.text_encoder(ids.to(device), mask.to(device)).last_hidden_state.to(dtype=dtype, device=device)
        emb = [u[:v] for u, v in zip(emb, seq_lens)]
        emb = torch.stack([torch.cat([u, u.new_zeros(max_sequence_length - u.size(0), u.size(1))]) for u in emb])
        _, seq_len, _ = emb.shape
        return emb.repeat(1, num_videos_per_prompt, 1).view(bsz * num_videos_per_prompt, seq_len, -1)

    def encode_prompt(self, prompt, negative_prompt=None, do_classifier_free_guidance=True, num_videos_per_prompt=1,
                      prompt_embeds=None, negative_prompt_embeds=None, max_sequence_length=226, device=None,
                      dtype=None):
        """reference :258-337"""
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
            if bsz != len(negative_prompt):
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch "
                                 f"size {bsz}.")
            negative_prompt_embeds = self._get_t5_prompt_embeds(negative_prompt, num_videos_per_prompt,
                                                                max_sequence_length, device, dtype)
        return prompt_embeds, negative_prompt_embeds

    def check_inputs(self, prompt, negative_prompt, image, height, width, prompt_embeds=None,
                     negative_prompt_embeds=None, image_embeds=None, callback_on_step_end_tensor_inputs=None,
                     guidance_scale_2=None):
        """reference :339-398 (same conditions, same error classes)"""
        import PIL.Image
        if image is not None and image_embeds is not None:
            raise ValueError("Cannot forward both `image` and `image_embeds`.")
        if image is None and image_embeds is None:
            raise ValueError("Provide either `image` or `prompt_embeds`. Cannot leave both `image` and `image_embeds` "
                             "undefined.")
        if image is not None and not isinstance(image, (torch.Tensor, PIL.Image.Image)):
            raise ValueError(f"`image` has to be of type `torch.Tensor` or `PIL.Image.Image` but is {type(image)}")
        if height % 16 != 0 or width % 16 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 16 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(
                k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        elif negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        elif prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        elif negative_prompt is not None and not isinstance(negative_prompt, (str, list)):
            raise ValueError(f"`negative_prompt` has to be of type `str` or `list` but is {type(negative_prompt)}")
        if self.config.boundary_ratio is None and guidance_scale_2 is not None:
            raise ValueError("`guidance_scale_2` is only supported when the pipeline's `boundary_ratio` is not None.")
        # Wan2.2-5B geometry: VAE x16 and patch 2 -> the token grid needs multiples of 32
        ps = self.transformer.config.patch_size
        if (height // self.vae_scale_factor_spatial) % ps[1] or (width // self.vae_scale_factor_spatial) % ps[2] \
                or height % self.vae_scale_factor_spatial or width % self.vae_scale_factor_spatial:
            raise ValueError(f"`height`/`width` ({height}x{width}) must be multiples of "
                             f"{self.vae_scale_factor_spatial * ps[1]} for this model (e.g. 704x1280, not 720x1280).")

    # ---- conditions (reference :400-553) ----
    def _norm_latents(self, z, dtype):
        cfgv = self.vae.config
        mean = torch.tensor(cfgv.latents_mean).view(1, cfgv.z_dim, 1, 1, 1).to(z.device, dtype)
        inv_std = 1.0 / torch.tensor(cfgv.latents_std).view(1, cfgv.z_dim, 1, 1, 1).to(z.device, dtype)
        return (z.to(dtype) - mean) * inv_std

    def prepare_latents(self, image, traj_tensor, ID_tensor, batch_size, num_channels_latents=16, height=480,
                        width=832, num_frames=81, dtype=None, device=None, generator=None, latents=None,
                        last_image=None):
        """reference :400-553 (Wan2.2 / expand_timesteps branch)"""
        return self._prepare_conditions(image, traj_tensor, ID_tensor, batch_size, num_channels_latents, height, width,
                                        num_frames, dtype, device, generator, latents, last_image)

    def _prepare_conditions(self, image, traj_tensor, ID_tensor, batch_size, num_channels_latents=16, height=480,
                            width=832, num_frames=81, dtype=None, device=None, generator=None, latents=None,
                            last_image=None):
        if last_image is not None:
            raise NotImplementedError("last_image is a Wan2.1 FLF2V feature, outside the FrameINO path")
        nlf = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        lh, lw = height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial
        shape = (batch_size, num_channels_latents, nlf, lh, lw)
        if latents is None:
            # diffusers randn_tensor: sample on the generator's device (CPU by default), then move
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            latents = latents.to(device=device, dtype=dtype)
        vdt = self.vae.dtype
        video_condition = image.unsqueeze(2).to(device=device, dtype=vdt)             # [B, 3, 1, H, W]
        cond = self.vae.encode(video_condition).latent_dist.mode().repeat(batch_size, 1, 1, 1, 1)
        cond = self._norm_latents(cond, dtype)
        traj = traj_tensor.to(device, dtype=vdt).unsqueeze(0).permute(0, 2, 1, 3, 4)  # [1, C, F, H, W]
        plan = getattr(self, "parallel", None)
        if plan is not None and plan.world > 1 and self.shard_vae_encode and hasattr(self.vae, "encode_slab") and traj.shape[2] > 1:
            # the one long encode of a call (the trajectory video): every rank a slab of the frame (parallel.sharded_vae_encode)
            from .parallel import sharded_vae_encode
            traj_post = sharded_vae_encode(self.vae, traj, plan.rank, plan.world).latent_dist
        else:
            traj_post = self.vae.encode(traj).latent_dist
        traj_latents = self._norm_latents(traj_post.mode(), dtype)
        traj_latents = traj_latents.contiguous().float()
        id_cond = None
        if ID_tensor is not None and ID_tensor.shape[2] != 0:
            ID_tensor = ID_tensor.to(device=device, dtype=vdt)
            ids = []
            for fi in range(ID_tensor.shape[2]):            # the reference supports exactly one (F6); N>=1 here
                z = self.vae.encode(ID_tensor[:, :, fi].unsqueeze(2)).latent_dist.mode().repeat(batch_size, 1, 1, 1, 1)
                ids.append(self._norm_latents(z, dtype))
            id_cond = torch.cat(ids, dim=2)
            traj_latents = torch.cat([traj_latents, torch.zeros_like(id_cond[:traj_latents.shape[0]])], dim=2)
        mask = torch.ones(1, 1, nlf, lh, lw, dtype=dtype, device=device)
        mask[:, :, 0] = 0
        return latents, cond, traj_latents, id_cond, mask

    # ---- the hot loop ----
    def _step(self, st):
        """One denoise step on static buffers `st` (graph-capturable: no host sync, no allocation-dependent shapes)."""
        tr, o = self.transformer, self.transformer.ops
        x = o.wan_model_input(st.lat, st.cond, st.idl, st.traj, tr.dtype, out=st.x)[None]
        rows = (st.t_rows, st.sel)

        live = {"live_rows": st.live_rows} if getattr(st, "live_rows", None) is not None else {}

        def fwd(name, emb):
            with tr.cache_context(name):
                return tr(hidden_states=x, timestep=None, encoder_hidden_states=emb, return_dict=False,
                          attention_kwargs=st.attention_kwargs, timestep_rows=rows, **live)[0][0]

        plan = getattr(self, "parallel", None)
        if plan is not None and plan.interleave and st.cfg:
            # both branches on this rank's token shard, advanced alternately block by block on two streams: the K|V
            # all-gather of one branch overlaps the other branch's compute (frameino_amd/parallel.py)
            if self._streams is None:
                self._streams = (torch.cuda.Stream(), torch.cuda.Stream()) if x.is_cuda else (None, None)
            main = torch.cuda.current_stream() if x.is_cuda else None
            gens, outs = [], [None, None]
            # the branches' kernels run on the two side streams.  Inside a hipGraph capture their collectives are issued on the
            # step's own stream (TokenShard.issue_stream: the only call pattern a capture of this step survives on this image);
            # an eager step issues them from the branch's stream, which keeps the two branches independent of each other
            capturing = bool(x.is_cuda and torch.cuda.is_current_stream_capturing())
            try:
                for shard in plan.shards:
                    shard.issue_stream = main if capturing else None
                for name, emb, shard in (("cond", st.pe, plan.shards[0]), ("uncond", st.ne, plan.shards[1])):
                    gens.append((name, tr.forward_steps(hidden_states=x, timestep=None, encoder_hidden_states=emb,
                                                         return_dict=False, attention_kwargs=st.attention_kwargs,
                                                         timestep_rows=rows, shard=shard)))
                for s_ in self._streams:
                    if s_ is not None:
                        s_.wait_stream(main)
                alive = [True, True]
                while any(alive):
                    for i, (name, g) in enumerate(gens):
                        if not alive[i]:
                            continue
                        ctx_stream = torch.cuda.stream(self._streams[i]) if self._streams[i] is not None else _null_ctx()
                        with ctx_stream, tr.cache_context(name):
                            try:
                                next(g)
                            except StopIteration as done:
                                outs[i], alive[i] = done.value[0][0], False
                for s_ in self._streams:
                    if s_ is not None:
                        main.wait_stream(s_)
            finally:
                for shard in plan.shards:
                    shard.issue_stream = None
            pc, pu = outs
        elif plan is not None and plan.cfg_ways == 2 and st.cfg:
            # CFG branches on two rank groups; one exchange of noise_pred per step (frameino_amd/parallel.py)
            mine = fwd("cond", st.pe) if plan.cfg_idx == 0 else fwd("uncond", st.ne)
            pc, pu = plan.exchange_cfg(mine)
        elif st.cfg and self.cfg_streams and getattr(tr, "parallel", None) is None:
            # the two CFG branches as two concurrent kernel streams: each branch's launches are 2.3-10.7 "rounds" of
            # workgroups over the 256 CUs, and the CUs that one branch's last partial round leaves idle take the
            # other branch's workgroups (same for the synchronized epilogue bursts)
            if self._streams is None:
                self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
            main = torch.cuda.current_stream()
            s1, s2 = self._streams
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                pc = fwd("cond", st.pe)
            with torch.cuda.stream(s2):
                pu = fwd("uncond", st.ne)
            main.wait_stream(s1)
            main.wait_stream(s2)
        elif st.cfg and self.batch_cfg and st.pe_ne is not None and getattr(tr, "parallel", None) is None:
            # both CFG branches as ONE batch-2 forward (rows of the same GEMMs: identical per-row arithmetic, twice
            # the tiles per launch, weights streamed once).  The reference makes two calls (:862-882).
            with tr.cache_context("cfg"):
                both = tr(hidden_states=x.expand(2, -1, -1, -1, -1), timestep=None, encoder_hidden_states=st.pe_ne,
                          return_dict=False, attention_kwargs=st.attention_kwargs, timestep_rows=rows, **live)[0]
            pc, pu = both[0], both[1]
        else:
            pc = fwd("cond", st.pe)
            pu = fwd("uncond", st.ne) if st.cfg else None
        if st.unipc is not None:
            # the sampler the released Wan2.2 folder ships: corrector + predictor + CFG in one pass
            o.cfg_unipc_step_(st.lat, *st.unipc, pc, pu, st.coef)
        else:
            o.cfg_euler_step_(st.lat, pc, pu, st.guidance, st.dt,
                              round_out=getattr(self.scheduler, "cast_output_to_model_dtype", True))

    def make_state(self, latents, condition, traj_latents, id_latent, first_frame_mask, prompt_embeds,
                   negative_prompt_embeds, guidance_scale, attention_kwargs=None):
        """Static device buffers of the loop (what a captured step reads and writes)."""
        tr, dev = self.transformer, latents.device
        if latents.shape[0] != 1:
            raise NotImplementedError("the Wan2.2 FrameINO path is batch 1 (SURVEY F7, Appendix C)")
        c, fg, lh, lw = latents.shape[1:]
        nid = 0 if id_latent is None else id_latent.shape[2]
        ps = tr.config.patch_size
        tok_per_frame = (lh // ps[1]) * (lw // ps[2])
        st = SimpleNamespace()
        st.lat = latents[0].float().contiguous().clone()
        st.cond = condition[0, :, :1].float().contiguous()
        st.idl = None if id_latent is None else id_latent[0].float().contiguous()
        st.traj = traj_latents[0].float().contiguous()
        st.x = torch.empty((2 * c, fg + nid, lh, lw), dtype=tr.dtype, device=dev)
        # per-token timestep = mask*t (:842): first-frame tokens -> row 0 (t=0), all others incl. ID tokens -> row 1
        sel = torch.ones((fg + nid) * tok_per_frame, dtype=torch.int32, device=dev)
        sel[:tok_per_frame] = (first_frame_mask[0, 0, 0, ::ps[1], ::ps[2]].flatten() != 0).to(torch.int32)
        st.sel = sel
        # Token rows whose prediction this loop READS (WanTransformer3DModel.forward(live_rows=)): the ID frames' predictions are
        # dropped (:884-885), and the first latent frame is taken from the condition in every model input (:829) and in the
        # returned latents (:913) whenever its mask is zero -- what the model predicts for it never reaches an output.
        first_dead = bool((first_frame_mask[0, 0, 0] == 0).all()) if fg > 1 else False
        lo_live, hi_live = (tok_per_frame if first_dead else 0), fg * tok_per_frame
        st.live_rows = (lo_live, hi_live) if (hi_live - lo_live) < (fg + nid) * tok_per_frame else None
        st.t_rows = torch.zeros(2, dtype=torch.float32, device=dev)
        st.dt = torch.zeros(1, dtype=torch.float32, device=dev)
        # UniPC multistep history (last corrected sample, two x0 predictions) + this step's coefficient row
        st.unipc = None
        if getattr(self.scheduler, "kind", "euler") == "unipc":
            st.unipc = tuple(torch.zeros_like(st.lat) for _ in range(3))
            st.coef = torch.zeros(10, dtype=torch.float32, device=dev)
        st.pe = prompt_embeds.to(tr.dtype)
        st.ne = None if negative_prompt_embeds is None else negative_prompt_embeds.to(tr.dtype)
        st.cfg = guidance_scale > 1 and st.ne is not None
        st.pe_ne = torch.cat([st.pe, st.ne], dim=0).contiguous() if st.cfg and st.pe.shape == st.ne.shape else None
        st.guidance = float(guidance_scale)
        st.attention_kwargs = attention_kwargs
        return st

    def denoise(self, latents, condition, traj_latents, id_latent, first_frame_mask, prompt_embeds,
                negative_prompt_embeds, guidance_scale, num_inference_steps, attention_kwargs=None,
                callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), timesteps_set=False):
        """reference :809-913.  Returns the final latents [B, C, F, h, w] fp32.  The loop state is one sample's (the app
        and the evaluation scripts run batch 1); a batch -- a list of prompts, num_videos_per_prompt > 1 -- runs sample by sample:
        the samples of the reference's batched loop never meet, every one sees the noise row `prepare_latents` drew for it."""
        if latents.shape[0] > 1:
            def row(t, i):
                return t if t is None or t.shape[0] == 1 else t[i:i + 1]
            outs = []
            for i in range(latents.shape[0]):
                cb = callback_on_step_end
                outs.append(self.denoise(latents[i:i + 1], row(condition, i), row(traj_latents, i), row(id_latent, i),
                                         row(first_frame_mask, i), row(prompt_embeds, i), row(negative_prompt_embeds, i),
                                         guidance_scale, num_inference_steps, attention_kwargs, cb,
                                         callback_on_step_end_tensor_inputs, timesteps_set=True if timesteps_set or i > 0 else False))
            return torch.cat(outs, dim=0)
        dev = latents.device
        if not timesteps_set:
            self.scheduler.set_timesteps(num_inference_steps, device=dev)
        timesteps = self.scheduler.timesteps
        st = self.make_state(latents, condition, traj_latents, id_latent, first_frame_mask, prompt_embeds,
                             negative_prompt_embeds, guidance_scale, attention_kwargs)
        if callback_on_step_end is not None:
            st.live_rows = None          # a callback sees the latents after every step, the first frame's Euler update included
        if st.unipc is not None:
            coefs = self.scheduler.coefs.to(dev).clone()
            coefs[:, 0] = st.guidance
        else:
            dts = self.scheduler.dts.to(dev)
        ts_dev = timesteps.to(dev).float()
        self._num_timesteps = len(timesteps)

        from .graph_step import StepGraph, capture_error_mode, groups_capturable
        stepper = StepGraph(lambda: self._step(st), self.use_hip_graph,
                            callback_on_step_end is None and st.lat.is_cuda
                            and (self.use_hip_graph is True or tr_default_procs(self.transformer))
                            and groups_capturable(getattr(self, "parallel", None), self.use_hip_graph is True),
                            len(timesteps), capture_error_mode(getattr(self, "parallel", None)))
        for i in range(len(timesteps)):
            if self._interrupt:
                continue
            self._current_timestep = timesteps[i]
            st.t_rows[1:2].copy_(ts_dev[i:i + 1])          # device-to-device: no host sync
            if st.unipc is not None:
                st.coef.copy_(coefs[i])
            else:
                st.dt.copy_(dts[i:i + 1])
            stepper.step()
            if callback_on_step_end is not None:
                latents = st.lat[None]
                loc = {"latents": latents, "prompt_embeds": st.pe, "negative_prompt_embeds": st.ne}
                out = callback_on_step_end(self, i, timesteps[i], {k: loc[k] for k in callback_on_step_end_tensor_inputs})
                if "latents" in out and out["latents"] is not latents:
                    st.lat.copy_(out["latents"][0])
                pe, ne = out.pop("prompt_embeds", st.pe), out.pop("negative_prompt_embeds", st.ne)
                if pe is not st.pe or ne is not st.ne:
                    st.pe, st.ne = pe, ne
                    st.pe_ne = torch.cat([pe, ne], dim=0).contiguous() if st.cfg and pe.shape == ne.shape else None
        self._current_timestep = None
        stepper.close()
        return (1 - first_frame_mask) * condition + first_frame_mask * st.lat[None]          # :913

    @torch.no_grad()
    def __call__(self, image, prompt=None, negative_prompt=None, traj_tensor=None, ID_tensor=None, height=480,
                 width=832, num_frames=81, num_inference_steps=50, guidance_scale=5.0, guidance_scale_2=None,
                 num_videos_per_prompt=1, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, image_embeds=None, last_image=None, output_type="np", return_dict=True,
                 attention_kwargs=None, callback_on_step_end=None, callback_on_step_end_tensor_inputs=["latents"],
                 max_sequence_length=512):
        if hasattr(callback_on_step_end, "tensor_inputs"):
            callback_on_step_end_tensor_inputs = callback_on_step_end.tensor_inputs
        self.check_inputs(prompt, negative_prompt, image, height, width, prompt_embeds, negative_prompt_embeds,
                          image_embeds, callback_on_step_end_tensor_inputs, guidance_scale_2)
        if num_frames % self.vae_scale_factor_temporal != 1:
            num_frames = num_frames // self.vae_scale_factor_temporal * self.vae_scale_factor_temporal + 1
        num_frames = max(num_frames, 1)
        self._guidance_scale, self._attention_kwargs = guidance_scale, attention_kwargs
        self._current_timestep, self._interrupt = None, False
        device = self._execution_device
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, negative_prompt, self.do_classifier_free_guidance, num_videos_per_prompt, prompt_embeds,
            negative_prompt_embeds, max_sequence_length, device)
        tdt = self.transformer.dtype
        prompt_embeds = prompt_embeds.to(device=device, dtype=tdt)
        if negative_prompt_embeds is not None:
            negative_prompt_embeds = negative_prompt_embeds.to(device=device, dtype=tdt)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        image = self.video_processor.preprocess(image, height=height, width=width).to(device, dtype=torch.float32)
        latents, condition, traj_latents, id_cond, mask = self._prepare_conditions(
            image, traj_tensor, ID_tensor, batch_size * num_videos_per_prompt, self.vae.config.z_dim, height, width,
            num_frames, torch.float32, device, generator, latents, last_image)
        latents = self.denoise(latents, condition, traj_latents, id_cond, mask, prompt_embeds, negative_prompt_embeds,
                               guidance_scale, num_inference_steps, attention_kwargs, callback_on_step_end,
                               callback_on_step_end_tensor_inputs, timesteps_set=True)
        if output_type != "latent":
            cfgv = self.vae.config
            lat = latents.to(self.vae.dtype)
            mean = torch.tensor(cfgv.latents_mean).view(1, cfgv.z_dim, 1, 1, 1).to(lat.device, lat.dtype)
            inv_std = 1.0 / torch.tensor(cfgv.latents_std).view(1, cfgv.z_dim, 1, 1, 1).to(lat.device, lat.dtype)
            lat = lat / inv_std + mean                                                # :917-925
            plan = getattr(self, "parallel", None)
            if (plan is not None and plan.world > 1 and self.shard_vae_decode and hasattr(self.vae, "decode_slab")
                    and lat.shape[0] == 1):
                # every rank holds the same latents: each decodes its slab of the frame, one all-gather of video rows
                from .parallel import sharded_vae_decode
                video = sharded_vae_decode(self.vae, lat, plan.rank, plan.world)
            else:
                video = self.vae.decode(lat, return_dict=False)[0]
            video = self.video_processor.postprocess_video(video, output_type=output_type)
        else:
            video = latents
        self.maybe_free_model_hooks()
        if not return_dict:
            return (video,)
        return WanPipelineOutput(frames=video)
