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
