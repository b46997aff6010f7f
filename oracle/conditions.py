"""Oracle restatement of the trajectory-video builder.  Test infrastructure.

Follows /root/reference/data_loader/video_dataset_motion.py:120-206 (`VideoDataset_Motion.prepare_traj_tensor`), the
call app.py:616-620 makes (original size == target size, so the cv2.resize at :169 is the identity).  The painting
loop is the reference's own numpy code path restated 1:1.  `cv2.filter2D` (:172) is OpenCV -- third-party and absent
offline (`import cv2` fails here), so the blur is restated from its documented semantics (correlation with the kernel,
anchor at the centre, BORDER_REFLECT_101, float32 result) as a direct sum: parity unpinned for the blur, and because
`.astype(np.uint8)` truncates, a float result that lands an ulp under an integer (white background = 255 x sum of
weights) can differ by one grey level from an OpenCV build that filters through its DFT path.
The Gaussian kernel itself (:29, utils/optical_flow_utils.py:168-219) IS pinned: tests/golden/traj_kernel.npz holds
the reference function's output."""
import numpy as np

# :32-34 (the reference appends 100 random colours after these nine; they are not reproducible and not restated)
ALL_COLOR_CODES = [(255, 0, 0), (255, 255, 0), (0, 255, 0), (0, 255, 255), (255, 0, 255), (0, 0, 255),
                   (128, 128, 128), (64, 224, 208), (233, 150, 122)]


def bivariate_gaussian(kernel_size=45, sig=3.0):
    """utils/optical_flow_utils.py:168-219, isotropic branch."""
    ax = np.arange(-kernel_size // 2 + 1.0, kernel_size // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    k = np.exp(-0.5 * (xx ** 2 + yy ** 2) / sig ** 2)
    return k / np.sum(k)


def _reflect101(i, n):
    if n == 1:
        return 0
    while i < 0 or i >= n:
        i = -i if i < 0 else 2 * (n - 1) - i
    return i


def paint_frames(full_pred_tracks, height, width, dot_radius):
    """:126-160 -> float32 [F, H, W, 3] canvases (0..255)."""
    colors = ALL_COLOR_CODES[:len(full_pred_tracks[0])]
    r = int(dot_radius * height / 384)                                  # :131
    frames = []
    for points_per_frame in full_pred_tracks:
        base = np.zeros((height, width, 3)).astype(np.float32)
        base.fill(255)
        for obj_idx, pts in enumerate(points_per_frame):
            for (hx, vy) in pts:
                if hx < 0 or hx >= width or vy < 0 or vy >= height:
                    continue
                v0, v1 = min(height, max(0, vy - r)), min(height, max(0, vy + r))
                h0, h1 = min(width, max(0, hx - r)), min(width, max(0, hx + r))
                base[v0:v1, h0:h1, :] = colors[obj_idx]
        frames.append(base)
    return np.stack(frames)


def prepare_traj_tensor(full_pred_tracks, height, width, dot_radius, kernel=None):
    """-> float32 [F, 3, H, W] in [-1, 1] (original size == target size)."""
    k = bivariate_gaussian() if kernel is None else kernel
    ks = k.shape[0]
    half = ks // 2
    canv = paint_frames(full_pred_tracks, height, width, dot_radius)    # [F, H, W, 3]
    rows = np.array([[_reflect101(y + d - half, height) for d in range(ks)] for y in range(height)])
    cols = np.array([[_reflect101(x + d - half, width) for d in range(ks)] for x in range(width)])
    # filter the "ink" 255 - v and convert back: the same filter (weights sum to 1), but white stays exactly 255
    # (255 x sum-of-weights lands an ulp above or below 255 by the luck of rounding, and astype(uint8) truncates)
    ink = 255.0 - canv.astype(np.float64)
    out = np.zeros_like(ink)
    for dy in range(ks):
        for dx in range(ks):
            out += k[dy, dx] * ink[:, rows[:, dy]][:, :, cols[:, dx]]
    q = (255.0 - out).astype(np.float32).astype(np.uint8).astype(np.float32)      # :172
    return np.transpose(q / 255.0 * 2.0 - 1.0, (0, 3, 1, 2)).astype(np.float32)   # :39-42, :187


# ------------------------------------------------------------------------------------------------ canvas / ID builders
def resize_area(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA) restated from its documented semantics ("resampling
    using pixel area relation"): destination pixel = overlap-weighted mean of the source pixels its footprint covers,
    float accumulate, saturate_cast<uchar> (round half to even).  OpenCV is third-party and absent offline: UNPINNED
    (its integer-factor fast path rounds half up, and its up-scaling branch uses its own coefficient rule)."""
    src = np.asarray(img, dtype=np.float64)
    h, w = src.shape[:2]

    def weights(n_src, n_dst):
        s = n_src / n_dst
        m = np.zeros((n_dst, n_src))
        for d in range(n_dst):
            lo, hi = d * s, min((d + 1) * s, n_src)
            for i in range(int(np.floor(lo)), min(n_src, int(np.ceil(hi)))):
                ov = min(hi, i + 1) - max(lo, i)
                if ov > 0:
                    m[d, i] = ov
        return m / m.sum(axis=1, keepdims=True)

    wy, wx = weights(h, out_h), weights(w, out_w)
    out = np.einsum("yh,hxc->yxc", wy, np.einsum("hwc,xw->hxc", src, wx, optimize=True), optimize=True)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def build_inference_canvas(first_frame, resized_h, resized_w, tl_h, tl_w, br_h, br_w):
    """app.py:270-350 (`inference_canvas`)."""
    eh, ew = resized_h + tl_h + br_h, resized_w + tl_w + br_w
    canvas = np.zeros((eh, ew, 3), dtype=np.uint8)
    canvas[tl_h:eh - br_h, tl_w:ew - br_w] = resize_area(first_frame, resized_h, resized_w)
    return canvas


def pad_id_reference(ref, canvas_h, canvas_w):
    """app.py:662-681 (after the SAM mask)."""
    rh, rw = ref.shape[:2]
    scale_h, scale_w = canvas_h / max(rh, rw), canvas_w / max(rh, rw)
    nh, nw = int(rh * scale_h), int(rw * scale_w)
    img = resize_area(ref, nh, nw)
    p1, q1 = (canvas_h - nh) // 2, (canvas_w - nw) // 2
    return np.pad(img, ((p1, canvas_h - nh - p1), (q1, canvas_w - nw - q1), (0, 0)), mode="constant", constant_values=0)
