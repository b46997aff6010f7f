"""Stand-in for diffusers.video_processor.VideoProcessor (third-party, restated):
PIL -> resize(lanczos) -> [0,1] -> 2x-1;  postprocess: x/2+0.5 clamp -> [B,F,H,W,C] numpy."""
import numpy as np
import PIL.Image
import torch


class VideoProcessor:
    def __init__(self, vae_scale_factor=8, do_resize=True):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        if isinstance(image, PIL.Image.Image):
            image = image.resize((width, height), resample=PIL.Image.LANCZOS)
            arr = np.array(image).astype(np.float32) / 255.0
            if arr.ndim == 2:
                arr = arr[..., None]
            t = torch.from_numpy(arr.transpose(2, 0, 1))[None]
            return 2.0 * t - 1.0
        if isinstance(image, torch.Tensor):
            t = image if image.ndim == 4 else image[None]
            return t if t.min() < 0 else 2.0 * t - 1.0
        raise ValueError(type(image))

    def postprocess_video(self, video, output_type="np"):
        outs = []
        for b in range(video.shape[0]):
            v = video[b].permute(1, 0, 2, 3)
            v = (v / 2 + 0.5).clamp(0, 1)
            if output_type == "np":
                outs.append(v.cpu().permute(0, 2, 3, 1).float().numpy())
            elif output_type == "pt":
                outs.append(v)
            else:
                raise ValueError(output_type)
        return np.stack(outs) if output_type == "np" else torch.stack(outs)
