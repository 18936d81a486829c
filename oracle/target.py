"""Oracle (test infrastructure only): keypoint -> Gaussian target maps, numpy.

T1 follows reference datasets/coco_dataset.py:185-250 (`COCOPoseDataset._generate_target`,
the generator train.py uses); T2 follows data/pose_transforms.py:385-457 (`GenerateTarget`,
dense variant).  Pinned by tests/golden/t1_target.npz and t2_dense_target.npz.
"""
import math

import numpy as np


def gaussian_patch(sigma: float) -> np.ndarray:
    """The (2*3σ+1)² float32 patch of coco_dataset.py:227-233.

    `size = 2*3σ+1` is a Python float (13.0 for σ=2, 10.0 for σ=1.5 -> 10 samples with the
    centre at 5.0, i.e. an asymmetric patch); arithmetic is float32, `np.exp` on float32.
    """
    size = 2 * (sigma * 3) + 1
    ax = np.arange(0, size, 1, np.float32)
    c = size // 2
    return np.exp(-((ax[None, :] - c) ** 2 + (ax[:, None] - c) ** 2) / (2 * sigma ** 2))


def patch_lut(sigma: float):
    """(lut, n, c): values of the patch indexed by integer d² = dx²+dy², its side and centre."""
    g = gaussian_patch(sigma)
    n = g.shape[0]
    c = int((2 * (sigma * 3) + 1) // 2)
    lut = np.zeros(2 * max(c, n - 1 - c) ** 2 + 1, np.float32)
    for j in range(n):
        for i in range(n):
            lut[(i - c) ** 2 + (j - c) ** 2] = g[j, i]
    return lut, n, c


def generate_target(keypoints, visible, input_size, heatmap_size, sigma):
    """One sample. keypoints (K,2) float32 input-px, visible (K,), sizes are (W,H).

    Returns target (K,Hh,Wh) float32 and weight (K,1) float32 (keeps COCO v=2 as 2.0).
    """
    keypoints = np.asarray(keypoints, np.float32)
    visible = np.asarray(visible, np.float32)
    K = keypoints.shape[0]
    wh, hh = int(heatmap_size[0]), int(heatmap_size[1])
    stride = np.asarray(input_size, np.float64) / np.asarray(heatmap_size, np.float64)
    reach = sigma * 3
    g = gaussian_patch(sigma)
    target = np.zeros((K, hh, wh), np.float32)
    weight = np.zeros((K, 1), np.float32)
    for k in range(K):
        weight[k, 0] = visible[k]
        if weight[k, 0] < 0.5:
            continue
        mx = float(keypoints[k, 0]) / stride[0]          # float32 value, float64 divide
        my = float(keypoints[k, 1]) / stride[1]
        # Python int() truncates toward zero (coco_dataset.py:219-220), it does not floor.
        x_lo, y_lo = math.trunc(mx - reach), math.trunc(my - reach)
        x_hi, y_hi = math.trunc(mx + reach + 1), math.trunc(my + reach + 1)
        if x_lo >= wh or y_lo >= hh or x_hi < 0 or y_hi < 0:
            weight[k, 0] = 0.0
            continue
        cols = range(max(0, x_lo), min(x_hi, wh))
        rows = range(max(0, y_lo), min(y_hi, hh))
        for y in rows:
            for x in cols:
                target[k, y, x] = g[y - y_lo, x - x_lo]
    return target, weight


def generate_target_batch(keypoints, visible, input_size, heatmap_size, sigma):
    ts, ws = zip(*(generate_target(k, v, input_size, heatmap_size, sigma) for k, v in zip(keypoints, visible)))
    return np.stack(ts), np.stack(ws)


def dense_target(keypoints, visible, input_size_hw, heatmap_size_hw, sigma):
    """T2: whole-map Gaussian at the sub-pixel centre (pose_transforms.py:396-455).

    Sizes are (H,W) here (the reference unpacks `heatmap_h, heatmap_w = heatmap_size`).
    weight = 1 for visible>0 and centre inside [0,W)x[0,H), else 0.
    """
    keypoints = np.asarray(keypoints, np.float32)
    K = keypoints.shape[0]
    hh, hw = heatmap_size_hw
    ih, iw = input_size_hw
    scaled = keypoints.copy()
    scaled[:, 0] *= hw / iw
    scaled[:, 1] *= hh / ih
    xs = np.arange(0, hw, 1, dtype=np.float32)[None, :]
    ys = np.arange(0, hh, 1, dtype=np.float32)[:, None]
    maps = np.zeros((K, hh, hw), np.float32)
    w = np.ones(K, np.float32)
    for k in range(K):
        cx, cy = scaled[k, 0], scaled[k, 1]
        if visible[k] > 0 and 0 <= cx < hw and 0 <= cy < hh:
            maps[k] = np.maximum(maps[k], np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma ** 2)))
        else:
            w[k] = 0.0
    return maps, w
