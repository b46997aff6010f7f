"""Stage-1 CogVideoX motion pipeline (reference pipelines/pipeline_cogvideox_i2v_motion.py; BASELINE config 1's shape
class): a `use_FrameIn=False` transformer, no identity frame, RoPE not extended -- model input
`[noisy | first-frame | trajectory]` on the channel axis (:771-790 there), one B=2 forward per step, guidance,
v-prediction DDIM update.  Shares every kernel and the loop with pipeline_cogvideox_i2v_motion_frameino.py."""
import torch

from .pipeline_cogvideox_i2v_motion_frameino import CogVideoXImageToVideoPipeline as _FrameINOPipeline
from .pipeline_cogvideox_i2v_motion_frameino import CogVideoXPipelineOutput          # noqa: F401  (re-export)


class CogVideoXImageToVideoPipeline(_FrameINOPipeline):
    extend_rope_by_first_frame = False            # the FrameINO pipeline's :834-839 does not exist in stage 1

    @torch.no_grad()
    def denoise(self, latents, image_latents, traj_latents, prompt_embeds, negative_prompt_embeds, guidance_scale=6.0,
                num_inference_steps=50, use_dynamic_cfg=False, image_rotary_emb=None, attention_kwargs=None,
                callback_on_step_end=None, generator=None):
        return super().denoise(latents, image_latents, traj_latents, None, prompt_embeds, negative_prompt_embeds,
                               guidance_scale, num_inference_steps, use_dynamic_cfg, image_rotary_emb,
                               attention_kwargs, callback_on_step_end, generator)
