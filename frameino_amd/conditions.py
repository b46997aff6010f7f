"""Condition builders in front of the denoising path, on the device.

`prepare_traj_tensor` mirrors `VideoDataset_Motion.prepare_traj_tensor`
(data_loader/video_dataset_motion.py:120-206) as app.py:616-620 calls it: point tracks -> per-frame white canvases with
coloured squares -> 45x45 Gaussian blur -> uint8 -> [-1, 1] `traj_tensor [F, 3, H, W]` (what the pipeline's
`traj_tensor=` argument takes).  Two HIP kernels (paint, separable blur + quantise) instead of numpy + OpenCV on the
host.  `original != target` sizes resize the canvases with torch's bicubic interpolation (the same a = -0.75 kernel
and half-pixel mapping as cv2.INTER_CUBIC, not bit-pinned: OpenCV is absent offline)."""
import numpy as np
import torch

from . import _lib

# video_dataset_motion.py:32-34 (the reference appends 100 unseeded random colours after these nine)
ALL_COLOR_CODES = [(255, 0, 0), (255, 255, 0), (0, 255, 0), (0, 255, 255), (255, 0, 255), (0, 0, 255),
                   (128, 128, 128), (64, 224, 208), (233, 150, 122)]


def gaussian_taps(kernel_size=45, sig=3.0):
    """1-D factor of the isotropic kernel of optical_flow_utils.py:197-219: outer(g, g) == bivariate_Gaussian(...)."""
    ax = np.arange(-kernel_size // 2 + 1.0, kernel_size // 2 + 1.0)
    g = np.exp(-0.5 * ax ** 2 / sig ** 2)
    return g / g.sum()


def prepare_traj_tensor(full_pred_tracks, original_height, original_width, dot_radius, target_width, target_height,
                        device="cuda", color_codes=None, kernel_size=45, sigma=3.0):
    """full_pred_tracks: [frames][objects][points] of (horizontal, vertical) integer pixel positions.
    Returns traj_tensor float32 [F, 3, target_height, target_width] in [-1, 1] on `device`."""
    frames = len(full_pred_tracks)
    colors = (color_codes or ALL_COLOR_CODES)[:len(full_pred_tracks[0])]
    radius = int(dot_radius * original_height / 384)                     # :131
    rows, offs = [], [0]
    for points_per_frame in full_pred_tracks:
        for obj_idx, pts in enumerate(points_per_frame):
            r, g, b = colors[obj_idx]
            rgb = int(r) | (int(g) << 8) | (int(b) << 16)
            rows.extend((int(x), int(y), rgb, 0) for (x, y) in pts)
        offs.append(len(rows))
    dev = torch.device(device)
    pts = torch.tensor(rows if rows else [(0, 0, 0, 0)], dtype=torch.int32).to(dev)
    off = torch.tensor(offs, dtype=torch.int32).to(dev)
    canvas = torch.empty((frames, 3, original_height, original_width), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    lib = _lib.lib()
    _lib.check(lib.fino_traj_paint(pts.data_ptr(), off.data_ptr(), canvas.data_ptr(), frames, original_height,
                                   original_width, radius, stream), "fino_traj_paint")
    if (target_height, target_width) != (original_height, original_width):
        canvas = torch.nn.functional.interpolate(canvas, size=(target_height, target_width), mode="bicubic",
                                                 align_corners=False).contiguous()      # :169 cv2.INTER_CUBIC
    taps = torch.tensor(gaussian_taps(kernel_size, sigma), dtype=torch.float32).to(dev)
    scratch, out = torch.empty_like(canvas), torch.empty_like(canvas)
    _lib.check(lib.fino_traj_blur_quantize(canvas.data_ptr(), scratch.data_ptr(), out.data_ptr(), taps.data_ptr(),
                                           kernel_size, frames * 3, target_height, target_width, stream),
               "fino_traj_blur_quantize")
    return out
