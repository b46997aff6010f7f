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
