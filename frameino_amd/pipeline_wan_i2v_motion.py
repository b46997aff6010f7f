"""Stage-1 Wan motion pipeline (reference pipelines/pipeline_wan_i2v_motion.py): the FrameINO pipeline without identity
reference frames -- same denoise loop (:757-870 there), `prepare_latents` returns four tensors (:490-492), `__call__`
has no `ID_tensor` (:541-568).  Every kernel and every host step is shared with
frameino_amd/pipeline_wan_i2v_motion_frameino.py; this module only restores the stage-1 signatures."""
import torch

from .pipeline_wan_i2v_motion_frameino import WanImageToVideoPipeline as _FrameINOPipeline
from .pipeline_wan_i2v_motion_frameino import WanPipelineOutput                      # noqa: F401  (re-export)


class WanImageToVideoPipeline(_FrameINOPipeline):
    def prepare_latents(self, image, traj_tensor, batch_size, num_channels_latents=16, height=480, width=832,
                        num_frames=81, dtype=None, device=None, generator=None, latents=None, last_image=None):
        latents, cond, traj_latents, _, mask = self._prepare_conditions(
            image, traj_tensor, None, batch_size, num_channels_latents, height, width, num_frames, dtype, device,
            generator, latents, last_image)
        return latents, cond, traj_latents, mask

    @torch.no_grad()
    def __call__(self, image, prompt=None, negative_prompt=None, traj_tensor=None, height=480, width=832,
                 num_frames=81, num_inference_steps=50, guidance_scale=5.0, guidance_scale_2=None,
                 num_videos_per_prompt=1, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, image_embeds=None, last_image=None, output_type="np", return_dict=True,
                 attention_kwargs=None, callback_on_step_end=None, callback_on_step_end_tensor_inputs=["latents"],
                 max_sequence_length=512):
        return _FrameINOPipeline.__call__(
            self, image, prompt=prompt, negative_prompt=negative_prompt, traj_tensor=traj_tensor, ID_tensor=None,
            height=height, width=width, num_frames=num_frames, num_inference_steps=num_inference_steps,
            guidance_scale=guidance_scale, guidance_scale_2=guidance_scale_2,
            num_videos_per_prompt=num_videos_per_prompt, generator=generator, latents=latents,
            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, image_embeds=image_embeds,
            last_image=last_image, output_type=output_type, return_dict=return_dict,
            attention_kwargs=attention_kwargs, callback_on_step_end=callback_on_step_end,
            callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs,
            max_sequence_length=max_sequence_length)
