"""Per-keypoint Gaussian target generation on the GPU (the file BASELINE.json names as data/generate_heatmap.py;
in the reference that file is empty and the live generator is datasets/coco_dataset.py:185-250).

`generate_target` is bit-exact with `COCOPoseDataset._generate_target` (float64 index arithmetic with C truncation,
values from a host-built LUT of the reference's own float32 numpy expression); `generate_dense_target` restates the
alternative `GenerateTarget` of data/pose_transforms.py:385-457.
"""
import torch

from .. import hipops


def generate_target(keypoints: torch.Tensor, keypoints_visible: torch.Tensor, input_size=(192, 256), heatmap_size=(48, 64),
                    sigma: float = 2.0):
    """keypoints (B,K,2) or (K,2) input-px, visible (B,K) or (K,) -> target (B,K,Hh,Wh), target_weight (B,K,1)."""
    single = keypoints.dim() == 2
    kp = keypoints[None] if single else keypoints
    vis = keypoints_visible[None] if single else keypoints_visible
    t, w = hipops.gaussian_target(kp.float(), vis.float(), input_size, heatmap_size, sigma)
    return (t[0], w[0]) if single else (t, w)


def generate_dense_target(keypoints, keypoints_visible, input_size_hw=(256, 256), heatmap_size_hw=(64, 64), sigma: float = 2.0):
    single = keypoints.dim() == 2
    kp = keypoints[None] if single else keypoints
    vis = keypoints_visible[None] if single else keypoints_visible
    h, w = hipops.dense_target(kp.float(), vis.float(), input_size_hw, heatmap_size_hw, sigma)
    return (h[0], w[0]) if single else (h, w)


class HeatmapGenerator:
    """Callable bound to a config: `gen(keypoints, visible) -> (target, target_weight)`."""

    def __init__(self, input_size=(192, 256), heatmap_size=(48, 64), sigma=2.0):
        self.input_size, self.heatmap_size, self.sigma = tuple(input_size), tuple(heatmap_size), float(sigma)

    def __call__(self, keypoints, keypoints_visible):
        return generate_target(keypoints, keypoints_visible, self.input_size, self.heatmap_size, self.sigma)
