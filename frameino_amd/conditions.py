"""Condition builders in front of the denoising path, on the device.

`prepare_traj_tensor` mirrors `VideoDataset_Motion.prepare_traj_tensor`
(data_loader/video_dataset_motion.py:120-206) as app.py:616-620 calls it: point tracks -> per-frame white canvases with
coloured squares -> 45x45 Gaussian blur -> uint8 -> [-1, 1] `traj_tensor [F, 3, H, W]` (what the pipeline's
`traj_tensor=` argument takes).  Two HIP kernels (paint, separable blur + quantise) instead of numpy + OpenCV on the
host.  `original != target` sizes resize the canvases with torch's bicubic interpolation (the same a = -0.75 kernel
and half-pixel mapping as cv2.INTER_CUBIC, not bit-pinned: OpenCV is absent offline).

`build_inference_canvas`, `prepare_id_tensor`, `tracks_from_trajectories` and `crop_unpadded` mirror the rest of what
app.py does between the UI and the pipeline call (:270-350, :487-518, :582-620, :634-695, :745-748): the first frame
area-resampled into the black unbounded canvas, the clicked trajectories resampled by arc length and scaled to canvas
pixels, the identity reference scaled, centred and zero-padded to the canvas, and the generated frames cropped back to
the original region.  SAM segmentation of the reference (:646-659) is a third-party model and stays with the caller:
pass the already-masked image."""
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


def _u8_image(img, device):
    t = torch.as_tensor(np.ascontiguousarray(img) if isinstance(img, np.ndarray) else img)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected an HWC uint8 image with 3 channels")
    return t.to(device).contiguous()


def resize_area_pad(img, region_hw, out_hw, offset_yx, fill=0, device="cuda"):
    """uint8 [H, W, 3] -> uint8 [out_h, out_w, 3]: `img` area-resampled (INTER_AREA) to `region_hw`, placed at
    `offset_yx`, `fill` elsewhere."""
    src = _u8_image(img, torch.device(device))
    out = torch.empty((out_hw[0], out_hw[1], 3), dtype=torch.uint8, device=src.device)
    _lib.check(_lib.lib().fino_resize_area_pad_u8(src.data_ptr(), out.data_ptr(), src.shape[0], src.shape[1], region_hw[0],
                                                 region_hw[1], out_hw[0], out_hw[1], offset_yx[0], offset_yx[1], fill,
                                                 torch.cuda.current_stream().cuda_stream), "fino_resize_area_pad_u8")
    return out


def to_unit_chw(img_u8):
    """uint8 [H, W, 3] (device) -> fp32 [3, H, W] in [-1, 1] (app.py:109-113 + :692)."""
    out = torch.empty((3, img_u8.shape[0], img_u8.shape[1]), dtype=torch.float32, device=img_u8.device)
    _lib.check(_lib.lib().fino_u8_hwc_to_chw_unit(img_u8.data_ptr(), out.data_ptr(), img_u8.shape[0], img_u8.shape[1],
                                                 torch.cuda.current_stream().cuda_stream), "fino_u8_hwc_to_chw_unit")
    return out


def build_inference_canvas(first_frame, resized_height, resized_width, top_left_height, top_left_width,
                           bottom_right_height, bottom_right_width, device="cuda"):
    """app.py::build_canvas (:270-350), the `inference_canvas` it returns: the first frame resized (INTER_AREA) to
    resized_height x resized_width inside a black canvas extended by the outside-region pads.  Same checks and
    messages (ValueError instead of gr.Error)."""
    eh = resized_height + top_left_height + bottom_right_height
    ew = resized_width + top_left_width + bottom_right_width
    if eh % 32 != 0:
        raise ValueError("The Height of resized_height + top_left_height + bottom_right_height must be divisible by 32!")
    if ew % 32 != 0:
        raise ValueError("The Width of resized_width + top_left_width + bottom_right_width must be divisible by 32!")
    return resize_area_pad(first_frame, (resized_height, resized_width), (eh, ew), (top_left_height, top_left_width), 0,
                           device)


def id_reference_geometry(ref_h, ref_w, canvas_height, canvas_width):
    """app.py:662-672: (new_h, new_w, pad_top, pad_left) of the scaled, centred identity reference."""
    scale_h = canvas_height / max(ref_h, ref_w)
    scale_w = canvas_width / max(ref_h, ref_w)
    new_h, new_w = int(ref_h * scale_h), int(ref_w * scale_w)
    return new_h, new_w, (canvas_height - new_h) // 2, (canvas_width - new_w) // 2


def prepare_id_tensor(reference_img, canvas_height, canvas_width, model_code_name="Wan", device="cuda"):
    """app.py:634-695 after the SAM mask: the (masked) reference scaled by canvas / max(ref_h, ref_w) per axis
    (INTER_AREA), zero-padded to the canvas around its centre, mapped to [-1, 1].  None -> the black placeholder
    (:683-685).  Returns [1, 3, 1, H, W] for Wan (:694-695), [3, H, W] for CogVideoX."""
    dev = torch.device(device)
    if reference_img is None:
        canvas = torch.zeros((canvas_height, canvas_width, 3), dtype=torch.uint8, device=dev)
    else:
        ref_h, ref_w = reference_img.shape[:2]
        new_h, new_w, top, left = id_reference_geometry(ref_h, ref_w, canvas_height, canvas_width)
        if new_h <= 0 or new_w <= 0:
            raise ValueError("identity reference collapses to an empty image at this canvas size")
        canvas = resize_area_pad(reference_img, (new_h, new_w), (canvas_height, canvas_width), (top, left), 0, dev)
    t = to_unit_chw(canvas)
    return t[None, :, None] if model_code_name == "Wan" else t


def sample_traj_by_length(points, num_samples):
    """app.py:487-518: `num_samples` points evenly spaced by arc length along the clicked polyline (host: a handful
    of points)."""
    pts = np.array(points, dtype=float)
    seg = pts[1:] - pts[:-1]
    seg_len = np.sqrt((seg ** 2).sum(axis=1))
    cum = np.cumsum(seg_len)
    target = np.linspace(0, cum[-1], num_samples)
    res = []
    for t in target:
        idx = np.searchsorted(cum, t)
        prev = 0.0 if idx == 0 else cum[idx - 1]
        ratio = (t - prev) / seg_len[idx]
        res.append(pts[idx] * (1 - ratio) + pts[idx + 1] * ratio)
    return np.array(res)


def tracks_from_trajectories(traj_lists, num_frames, canvas_height, canvas_width, uniform_height, uniform_width):
    """app.py:582-612: [instances][trajectories][clicked (x, y) on the uniform_height x uniform_width board] ->
    full_pred_tracks [frames][instances][points] in canvas pixels (the argument of `prepare_traj_tensor`)."""
    tracks = [[] for _ in range(num_frames)]
    for traj_list_per_object in traj_lists:
        for traj_idx, single in enumerate(traj_list_per_object):
            if len(single) < 2:
                raise ValueError("One of the trajectory provided is too short!")
            for f, (rx, ry) in enumerate(sample_traj_by_length(single, num_frames)):
                if traj_idx == 0:
                    tracks[f].append([])
                tracks[f][-1].append((int(rx * canvas_width / uniform_width), int(ry * canvas_height / uniform_height)))
    return tracks


def crop_unpadded(frames, top_left_height, top_left_width, bottom_right_height, bottom_right_width):
    """app.py:741-748: frames [F, H, W, 3] in [0, 1] -> uint8 frames of the original (un-extended) region."""
    f = torch.as_tensor(frames)
    y1, x1 = f.shape[1] - bottom_right_height, f.shape[2] - bottom_right_width
    return (f[:, top_left_height:y1, top_left_width:x1] * 255).to(torch.uint8)     # np.uint8(x * 255): truncation
