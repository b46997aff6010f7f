"""Trajectory-video condition builder: the oracle's Gaussian kernel against the reference function's
recorded output, and the HIP builder against the oracle."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _tracks(frames, height, width, seed=0):
    rng = np.random.default_rng(seed)
    tracks = []
    for f in range(frames):
        objs = []
        for o in range(3):
            n = 4 + o
            xs = rng.integers(-5, width + 5, n)              # some points fall outside the canvas (:149-150)
            ys = rng.integers(-5, height + 5, n)
            objs.append([(int(x), int(y)) for x, y in zip(xs, ys)])
        tracks.append(objs)
    tracks[0][0].append((0, 0))                               # corner: clipped square
    tracks[0][1].append((width - 1, height - 1))
    return tracks


def test_gaussian_kernel_matches_reference_function():
    from frameino_amd.conditions import gaussian_taps
    from oracle.conditions import bivariate_gaussian
    ref = np.load(os.path.join(GOLD, "traj_kernel.npz"))["kernel"]
    np.testing.assert_allclose(bivariate_gaussian(45, 3.0), ref, rtol=1e-13, atol=0)
    g = gaussian_taps(45, 3.0)
    np.testing.assert_allclose(np.outer(g, g), ref, rtol=1e-12, atol=1e-30)     # the separable factor the kernels use


def test_oracle_painting_semantics():
    from oracle.conditions import ALL_COLOR_CODES, paint_frames
    tr = [[[(10, 12)], [(11, 12)]]]                          # object 1 painted after object 0: later paint wins
    c = paint_frames(tr, 48, 64, dot_radius=16)              # r = int(16 * 48 / 384) = 2
    assert tuple(c[0, 12, 10]) == ALL_COLOR_CODES[1] and tuple(c[0, 12, 8]) == ALL_COLOR_CODES[0]
    assert tuple(c[0, 12, 13]) == (255, 255, 255) and tuple(c[0, 9, 10]) == (255, 255, 255)   # [y-r, y+r) half-open
    assert tuple(c[0, 13, 12]) == ALL_COLOR_CODES[1]


@pytest.mark.gpu
def test_hip_builder_vs_oracle():
    from frameino_amd.conditions import prepare_traj_tensor
    from oracle.conditions import paint_frames, prepare_traj_tensor as oracle_traj
    frames, h, w, dot = 3, 96, 128, 24
    tr = _tracks(frames, h, w)
    out = prepare_traj_tensor(tr, h, w, dot, w, h, device="cuda")
    assert out.shape == (frames, 3, h, w) and out.dtype == torch.float32
    ref = oracle_traj(tr, h, w, dot)
    # grey levels: identical except where the fp32 sum lands within an ulp of an integer (truncation to uint8)
    lv = torch.round((out.cpu() + 1) * 127.5).numpy()
    lr = np.round((ref + 1) * 127.5)
    assert np.abs(lv - lr).max() <= 1 and (lv != lr).mean() < 0.02, (np.abs(lv - lr).max(), (lv != lr).mean())
    # painting alone is exact: blur with a 1-tap identity kernel
    ident = prepare_traj_tensor(tr, h, w, dot, w, h, device="cuda", kernel_size=1, sigma=1.0)
    canv = np.transpose(paint_frames(tr, h, w, dot), (0, 3, 1, 2)) / 255.0 * 2.0 - 1.0
    np.testing.assert_allclose(ident.cpu().numpy(), canv.astype(np.float32), atol=1e-6)


@pytest.mark.gpu
def test_hip_builder_app_resolution_runs():
    """49 frames at 704x1280 (the app's canvas): shape / range / white background."""
    from frameino_amd.conditions import prepare_traj_tensor
    tr = _tracks(49, 704, 1280, seed=1)
    out = prepare_traj_tensor(tr, 704, 1280, 6, 1280, 704, device="cuda")
    assert out.shape == (49, 3, 704, 1280)
    assert out.min().item() >= -1.0 and out.max().item() <= 1.0
    assert (out > 0.98).float().mean().item() > 0.9        # mostly white background (254 or 255 after truncation)


# ---------------------------------------------------------------------------------------- canvas / ID builders (app.py)
def test_oracle_area_resize_hand_computed_cases():
    """INTER_AREA's pixel-area relation on cases small enough to do by hand."""
    from oracle.conditions import resize_area
    img = np.zeros((2, 4, 3), dtype=np.uint8)
    img[..., 0] = [[10, 20, 30, 40], [50, 60, 70, 80]]
    out = resize_area(img, 1, 2)                               # 2x2 boxes: (10+20+50+60)/4 = 35, (30+40+70+80)/4 = 55
    assert out[0, :, 0].tolist() == [35, 55]
    row = np.zeros((1, 3, 3), dtype=np.uint8)
    row[0, :, 0] = [0, 90, 30]
    out = resize_area(row, 1, 2)                               # footprints [0,1.5), [1.5,3): (0 + 45)/1.5 = 30, (45+30)/1.5 = 50
    assert out[0, :, 0].tolist() == [30, 50]
    assert np.array_equal(resize_area(img, 2, 4), img)         # identity
    # constant images stay constant under any scale (weights sum to one), incl. up-scaling
    c = np.full((5, 7, 3), 137, dtype=np.uint8)
    assert (resize_area(c, 3, 4) == 137).all() and (resize_area(c, 9, 11) == 137).all()


def test_host_builders_cpu():
    from frameino_amd.conditions import (crop_unpadded, id_reference_geometry, sample_traj_by_length,
                                         tracks_from_trajectories)
    # arc-length sampling: an L-shaped polyline of length 10 + 10, 5 samples -> every 5 units
    pts = sample_traj_by_length([(0, 0), (10, 0), (10, 10)], 5)
    np.testing.assert_allclose(pts, [[0, 0], [5, 0], [10, 0], [10, 5], [10, 10]], atol=1e-9)
    tr = tracks_from_trajectories([[[(0, 0), (100, 0)], [(0, 50), (100, 50)]], [[(10, 10), (10, 90)]]], 3, 704, 1280,
                                  480, 720)
    assert len(tr) == 3 and len(tr[0]) == 2 and len(tr[0][0]) == 2 and len(tr[0][1]) == 1
    assert tr[2][0][0] == (int(100 * 1280 / 720), 0) and tr[1][1][0] == (int(10 * 1280 / 720), int(50 * 704 / 480))
    with pytest.raises(ValueError, match="too short"):
        tracks_from_trajectories([[[(1, 1)]]], 3, 64, 64, 64, 64)
    assert id_reference_geometry(300, 600, 704, 1280) == (352, 1280, 176, 0)      # app.py:662-672
    fr = torch.rand(2, 64, 96, 3)
    c = crop_unpadded(fr, 8, 16, 24, 32)
    assert c.shape == (2, 32, 48, 3) and c.dtype == torch.uint8
    assert torch.equal(c, (fr[:, 8:40, 16:64] * 255).to(torch.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("src_hw,geom", [((90, 160), (64, 96, 16, 32, 16, 0)), ((480, 832), (448, 704, 0, 64, 32, 64)),
                                          ((37, 53), (64, 64, 0, 0, 0, 0))])
def test_inference_canvas_vs_oracle(src_hw, geom):
    """app.py::build_canvas: first frame area-resampled into the black unbounded canvas (down- and up-scaling)."""
    from frameino_amd.conditions import build_inference_canvas
    from oracle import conditions as O
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, src_hw + (3,), dtype=np.uint8)
    out = build_inference_canvas(img, *geom).cpu().numpy()
    ref = O.build_inference_canvas(img, *geom)
    assert out.shape == ref.shape
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3              # fp32 vs fp64 accumulation at exact .5 ties
    rh, rw, tl_h, tl_w, br_h, br_w = geom
    assert (out[:tl_h] == 0).all() and (out[:, :tl_w] == 0).all() and (out[tl_h + rh:] == 0).all()
    with pytest.raises(ValueError, match="divisible by 32"):
        build_inference_canvas(img, 60, 96, 0, 0, 0, 0)


@pytest.mark.gpu
def test_id_tensor_vs_oracle():
    from frameino_amd.conditions import prepare_id_tensor
    from oracle import conditions as O
    rng = np.random.default_rng(2)
    ref_img = rng.integers(0, 256, (150, 90, 3), dtype=np.uint8)
    t = prepare_id_tensor(ref_img, 128, 192)
    assert t.shape == (1, 3, 1, 128, 192) and t.dtype == torch.float32
    exp = O.pad_id_reference(ref_img, 128, 192).astype(np.float32) / 255.0 * 2.0 - 1.0
    got = t[0, :, 0].permute(1, 2, 0).cpu().numpy()
    assert np.abs(got - exp).max() <= 2.0 / 255 + 1e-6 and (np.abs(got - exp) > 1e-6).mean() < 5e-3
    # reference absent: the all-black placeholder = -1 everywhere (app.py:683-685)
    z = prepare_id_tensor(None, 64, 96, model_code_name="CogVideoX")
    assert z.shape == (3, 64, 96) and (z == -1).all()


def test_bicubic_resize_of_the_trajectory_canvas_hand_computed():
    """`prepare_traj_tensor` resizes the painted canvases with torch's bicubic interpolation when original != target
    size (the reference: cv2.resize INTER_CUBIC, data_loader/video_dataset_motion.py:169).  Both use the Keys cubic
    with a = -0.75 on half-pixel centres; here the torch op is checked against that formula computed by hand on a
    1-D ramp with a step (2x up-scaling: source coordinate = (dst + 0.5) / 2 - 0.5, taps at floor - 1 .. floor + 2,
    borders clamped)."""
    def keys(x, a=-0.75):
        x = abs(x)
        if x <= 1:
            return (a + 2) * x ** 3 - (a + 3) * x ** 2 + 1
        if x < 2:
            return a * x ** 3 - 5 * a * x ** 2 + 8 * a * x - 4 * a
        return 0.0

    src = [255.0, 255.0, 0.0, 255.0, 128.0, 255.0]
    n = len(src)
    exp = []
    for d in range(2 * n):
        s = (d + 0.5) / 2 - 0.5
        f = int(np.floor(s))
        exp.append(sum(keys(s - (f + k)) * src[min(max(f + k, 0), n - 1)] for k in (-1, 0, 1, 2)))
    t = torch.tensor(src).view(1, 1, 1, n).repeat(1, 1, 3, 1)
    out = torch.nn.functional.interpolate(t, size=(3, 2 * n), mode="bicubic", align_corners=False)[0, 0, 1]
    np.testing.assert_allclose(out.numpy(), np.array(exp), rtol=0, atol=1e-3)
    assert out.max() > 255.0                                    # cubic overshoot next to the step: not clamped here,
    # the uint8 truncation at the end of prepare_traj_tensor (:172) happens after the blur


def test_blur_and_area_resampling_against_independent_library_implementations():
    """OpenCV is absent offline, so `cv2.filter2D(BORDER_REFLECT_101)` and `cv2.resize(INTER_AREA)` are restated from their
    documented semantics (oracle/conditions.py: parity with OpenCV itself stays unpinned).  Two INDEPENDENT implementations
    of the same documented operations are installed, though: scipy.ndimage.correlate with mode="mirror" (reflection without
    repeating the edge sample = REFLECT_101) and PIL's BOX filter (pixel-area average).  The restatement must agree with
    them -- which rules out a private misreading of the border rule, the kernel orientation or the area weights."""
    import PIL.Image
    from scipy import ndimage
    from oracle import conditions as C
    rng = np.random.default_rng(3)
    # (a) the 45-tap Gaussian blur of the trajectory canvas, before the uint8 truncation
    k = C.bivariate_gaussian()
    img = rng.uniform(0, 255, (37, 53)).astype(np.float64)
    half = k.shape[0] // 2
    rows = np.array([[C._reflect101(y + d - half, img.shape[0]) for d in range(k.shape[0])] for y in range(img.shape[0])])
    cols = np.array([[C._reflect101(x + d - half, img.shape[1]) for d in range(k.shape[1])] for x in range(img.shape[1])])
    mine = np.zeros_like(img)
    for dy in range(k.shape[0]):
        for dx in range(k.shape[1]):
            mine += k[dy, dx] * img[rows[:, dy]][:, cols[:, dx]]
    ref = ndimage.correlate(img, k, mode="mirror")
    assert np.abs(mine - ref).max() < 1e-9
    # an asymmetric kernel tells correlation (filter2D) from convolution and x from y
    ka = rng.uniform(0, 1, (5, 7))
    ka /= ka.sum()
    r5 = np.array([[C._reflect101(y + d - 2, 37) for d in range(5)] for y in range(37)])
    c7 = np.array([[C._reflect101(x + d - 3, 53) for d in range(7)] for x in range(53)])
    mine = sum(ka[dy, dx] * img[r5[:, dy]][:, c7[:, dx]] for dy in range(5) for dx in range(7))
    assert np.abs(mine - ndimage.correlate(img, ka, mode="mirror")).max() < 1e-9
    # (b) pixel-area resampling at integer shrink factors against PIL's BOX filter run on FLOAT planes (its uint8 path rounds
    # between its horizontal and vertical passes): the restatement's uint8 result must be a correct rounding of the
    # independent float result everywhere.  (At non-integer factors PIL's BOX weighs a source pixel 0 or 1 by its CENTRE,
    # not by the overlapped area, so it is no reference there: those cases stay hand-computed, above.)
    src = rng.integers(0, 256, (96, 120, 3), dtype=np.uint8)
    for oh, ow in ((48, 60), (32, 40), (24, 30), (48, 40)):
        mine = C.resize_area(src, oh, ow).astype(np.float64)
        for ch in range(3):
            plane = PIL.Image.fromarray(src[:, :, ch].astype(np.float32), mode="F")
            box = np.asarray(plane.resize((ow, oh), resample=PIL.Image.BOX)).astype(np.float64)
            d = np.abs(mine[:, :, ch] - box)
            assert d.max() <= 0.5 + 2e-3, (oh, ow, ch, d.max())
