"""Fusion head: heatmap + offset-regression + variance branches, its decoder and its loss
(drop-in for the reference's models/fusion_head.py).

Everything numerical runs in libposekernels: `decode` is one kernel per batch (no Python B x K loop, no .item()),
`FusionPoseLoss` is a fused forward (3 launches) + a hand-derived backward (1 launch).
"""
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import hipops
from .. import dispatch as nnops
from ._blocks import conv


class SoftArgmax2D(nn.Module):
    """Soft-argmax coordinates + raw-max scores (fusion_head.py:24-71), differentiable (coordinates through the softmax, scores to the
    first maximum: pk_softargmax_bwd)."""

    def __init__(self, beta: float = 1.0):
        super().__init__()
        self.beta = beta

    def forward(self, heatmaps: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        h = heatmaps.float()
        if torch.is_grad_enabled() and h.requires_grad:
            co, sc = hipops.softargmax(h)
            if self.beta != 1.0:  # coordinates from softmax(beta * H), scores stay the raw maxima (fusion_head.py:50-69)
                co = hipops.softargmax(h * self.beta)[0]
            return co, sc
        one = torch.full((1,), 40.0, device=heatmaps.device)          # sigmoid(40) == 1: pure global soft-argmax
        co, sc = hipops.softargmax_refine_decode(h, None, one, None, radius=0)
        if self.beta != 1.0:
            co, _ = hipops.softargmax_refine_decode((h * self.beta).contiguous(), None, one, None, radius=0)
        return co, sc


class LocalGaussianRefinement(nn.Module):
    """Softmax-weighted centroid of the clipped (2r+1)^2 patch around round(coarse_coords) (fusion_head.py:74-128), one kernel for the
    batch instead of the reference's B x K Python loop with .item(); like the reference's, the result carries no gradient."""

    def __init__(self, local_radius: int = 2):
        super().__init__()
        self.local_radius = local_radius

    @torch.no_grad()
    def forward(self, heatmaps: torch.Tensor, coarse_coords: torch.Tensor) -> torch.Tensor:
        return hipops.local_gaussian_refine(heatmaps.float(), coarse_coords.float(), self.local_radius)


class SubPixelRefinement(nn.Module):
    """alpha * global soft-argmax + (1 - alpha) * local centroid (fusion_head.py:131-172); the blend happens inside the decode kernel."""

    def __init__(self, beta: float = 1.0, local_radius: int = 2, fusion_alpha: float = 0.5):
        super().__init__()
        self.local_radius = local_radius
        self.soft_argmax = SoftArgmax2D(beta=beta)
        self.local_refine = LocalGaussianRefinement(local_radius=local_radius)
        self.alpha = nn.Parameter(torch.tensor(fusion_alpha))

    def forward(self, heatmaps):
        return hipops.softargmax_refine_decode(heatmaps.float(), None, self.alpha.detach(), None, self.local_radius)


def _branch(cin, hidden, cout):
    """[conv3x3, BN, ReLU, conv1x1+bias] with the reference's Sequential indices 0,1,(2),3."""
    seq = nn.ModuleDict({"0": conv(cin, hidden, 3), "1": nn.BatchNorm2d(hidden), "3": nn.Conv2d(hidden, cout, 1)})
    return seq


class HeatmapRegressionHead(nn.Module):
    def __init__(self, in_channels: int, num_keypoints: int = 17, hidden_dim: int = 256, use_subpixel_refinement: bool = True):
        super().__init__()
        self.in_channels, self.num_keypoints, self.use_subpixel_refinement = in_channels, num_keypoints, use_subpixel_refinement
        self.shared_layers = nn.ModuleDict({"0": conv(in_channels, hidden_dim, 3), "1": nn.BatchNorm2d(hidden_dim),
                                            "3": conv(hidden_dim, hidden_dim, 3), "4": nn.BatchNorm2d(hidden_dim)})
        self.heatmap_branch = _branch(hidden_dim, hidden_dim, num_keypoints)
        self.offset_branch = _branch(hidden_dim, hidden_dim, num_keypoints * 2)
        self.variance_branch = _branch(hidden_dim, hidden_dim // 2, num_keypoints)
        if use_subpixel_refinement:
            self.subpixel_refine = SubPixelRefinement(1.0, 2, 0.5)
        self.fusion_weight = nn.Parameter(torch.tensor(0.5))
        for m in self.modules():                                    # fusion_head.py:268-276
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        with nnops.scope(self):
            return self._forward(x)

    def _forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        tr = self.training
        x = nnops.from_public(x)
        s = self.shared_layers
        f = nnops.conv_bn_act(x, s["0"], s["1"], True, None, tr)
        f = nnops.conv_bn_act(f, s["3"], s["4"], True, None, tr)

        def make(br, softplus=False):
            def run(ins):
                t = nnops.conv_bn_act(ins[0], br["0"], br["1"], True, None, tr)
                return nnops.head_out(t, br["3"], softplus)
            return run

        # the three branches are independent: on concurrent streams their small BatchNorm kernels (finalize, partial sums: a
        # handful of workgroups each) hide under another branch's convolution instead of leaving the GPU idle
        # (measured in round 4: the three branches one after the other on one stream cost 0.35 ms per step -- DESIGN.md section 4)
        heatmaps, offsets, variances = nnops.parallel([make(self.heatmap_branch), make(self.offset_branch), make(self.variance_branch, True)],
                                                      [[f], [f], [f]])
        B, _, H, W = offsets.shape
        return {"heatmaps": heatmaps, "offsets": offsets.view(B, self.num_keypoints, 2, H, W),
                "variances": variances, "fusion_weight": torch.sigmoid(self.fusion_weight)}

    @torch.no_grad()
    def decode(self, outputs: Dict[str, torch.Tensor], apply_offset: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> keypoints (B,K,2) in heatmap pixels, scores (B,K) (fusion_head.py:309-365)."""
        hm = outputs["heatmaps"].float()
        if self.use_subpixel_refinement:
            alpha, radius = self.subpixel_refine.alpha.detach(), self.subpixel_refine.local_radius
        else:
            alpha, radius = torch.full((1,), 40.0, device=hm.device), 0
        if not apply_offset:
            return hipops.softargmax_refine_decode(hm, None, alpha, None, radius)
        fw = outputs["fusion_weight"].detach().float().reshape(1)
        raw = torch.log(fw / (1 - fw))                               # kernel applies sigmoid to the raw parameter
        return hipops.softargmax_refine_decode(hm, outputs["offsets"].float(), alpha, raw, radius)


_TERM = {"heatmap": 0, "offset": 1, "peak": 2, "variance": 3, "overlap": 4, "shape": 5}


def _lam(dev, thr=0.5, **terms):
    v = [0.0] * 6 + [float(thr)]
    for k, w in terms.items():
        v[_TERM[k]] = float(w)
    return torch.tensor(v, dtype=torch.float32, device=dev)


class GaussianDistributionConstraint(nn.Module):
    """Variance-alignment, spatial-overlap and distribution-shape terms (fusion_head.py:372-575) on ANY coordinates, each a binding of
    the fused loss kernels (pk_fusion_terms_fwd / _bwd: one term selected by its lambda); gradients flow to the heatmaps, the
    coordinates and the predicted variances as in the reference."""
    SKELETON = [(0, 1), (0, 2), (1, 3), (2, 4), (5, 6), (5, 7), (7, 9), (6, 8), (8, 10), (5, 11), (6, 12), (11, 12),
                (11, 13), (13, 15), (12, 14), (14, 16)]

    def __init__(self, target_sigma: float = 2.0, overlap_threshold: float = 0.5):
        super().__init__()
        self.target_sigma, self.overlap_threshold = target_sigma, overlap_threshold

    def _terms(self, heatmaps, coords, target_weight, pred_variances, **terms):
        return hipops.fusion_terms(_lam(heatmaps.device, self.overlap_threshold, **terms), hm=heatmaps, var=pred_variances, coords=coords,
                                   weight=target_weight, sigma_t=self.target_sigma)

    def compute_heatmap_variance(self, heatmaps: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
        """sigma (B,K) = sqrt(second moment of relu(H)/sum about coords + 1e-8) (fusion_head.py:405-448)."""
        return self._terms(heatmaps, coords, None, None)[1]

    def variance_alignment_loss(self, heatmaps, coords, target_weight, pred_variances: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._terms(heatmaps, coords, target_weight, pred_variances, variance=1.0)[0][3]

    def spatial_overlap_loss(self, heatmaps, target_weight) -> torch.Tensor:
        return self._terms(heatmaps, None, target_weight, None, overlap=1.0)[0][4]

    def distribution_shape_loss(self, heatmaps, target_weight) -> torch.Tensor:
        return self._terms(heatmaps, None, target_weight, None, shape=1.0)[0][5]

    def forward(self, heatmaps, coords, target_weight, pred_variances: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        v = self._terms(heatmaps, coords, target_weight, pred_variances, variance=1.0, overlap=1.0, shape=1.0)[0]
        return {"variance_loss": v[3], "overlap_loss": v[4], "shape_loss": v[5]}


class FusionPoseLoss(nn.Module):
    NAMES = ("heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss", "shape_loss", "total_loss")

    def __init__(self, heatmap_weight: float = 1.0, offset_weight: float = 1.0, peak_weight: float = 0.5,
                 variance_weight: float = 0.1, overlap_weight: float = 0.05, shape_weight: float = 0.05,
                 target_sigma: float = 2.0, use_target_weight: bool = True):
        super().__init__()
        self.heatmap_weight, self.offset_weight, self.peak_weight = heatmap_weight, offset_weight, peak_weight
        self.variance_weight, self.overlap_weight, self.shape_weight = variance_weight, overlap_weight, shape_weight
        self.use_target_weight = use_target_weight
        self.gaussian_constraint = GaussianDistributionConstraint(target_sigma)
        self.soft_argmax = SoftArgmax2D()
        self.register_buffer("_lambdas", torch.tensor([heatmap_weight, offset_weight, peak_weight, variance_weight, overlap_weight, shape_weight,
                                                       self.gaussian_constraint.overlap_threshold], dtype=torch.float32), persistent=False)

    # the three term methods of the reference (fusion_head.py:637-743), each on its own inputs
    def heatmap_loss(self, pred, target, weight) -> torch.Tensor:
        return hipops.fusion_terms(_lam(pred.device, heatmap=1.0), hm=pred, target=target, weight=weight,
                                   use_target_weight=self.use_target_weight)[0][0]

    def offset_loss(self, pred_offsets, pred_coords, gt_coords, weight, input_size, heatmap_size) -> torch.Tensor:
        """SmoothL1(offsets sampled at pred_coords, gt * (W/in_w, H/in_h) - pred_coords); heatmap_size = (H, W) (fusion_head.py:659-712)."""
        B, K = pred_coords.shape[:2]
        H, W = heatmap_size
        return hipops.fusion_terms(_lam(pred_offsets.device, offset=1.0), off=pred_offsets, coords=pred_coords, gt=gt_coords, weight=weight,
                                   input_size=input_size, use_target_weight=self.use_target_weight, shape=(B, K, H, W))[0][1]

    def peak_localization_loss(self, pred_coords, gt_coords, weight, input_size, heatmap_size) -> torch.Tensor:
        B, K = pred_coords.shape[:2]
        H, W = heatmap_size
        return hipops.fusion_terms(_lam(pred_coords.device, peak=1.0), coords=pred_coords, gt=gt_coords, weight=weight, input_size=input_size,
                                   use_target_weight=self.use_target_weight, shape=(B, K, H, W))[0][2]

    def forward(self, outputs, target_heatmaps, target_weight, gt_keypoints, input_size=(192, 256), heatmap_size=(48, 64)):
        vals = hipops.fusion_loss(outputs["heatmaps"], outputs["offsets"], outputs["variances"], target_heatmaps.float(),
                                  target_weight.float(), gt_keypoints.float(), input_size, self.gaussian_constraint.target_sigma,
                                  self._lambdas, self.use_target_weight)
        return {n: vals[i] for i, n in enumerate(self.NAMES)}


def build_fusion_head(in_channels: int, num_keypoints: int = 17, hidden_dim: int = 256) -> HeatmapRegressionHead:
    return HeatmapRegressionHead(in_channels, num_keypoints, hidden_dim, True)


def build_fusion_loss(target_sigma: float = 2.0, heatmap_weight: float = 1.0, offset_weight: float = 1.0,
                      variance_weight: float = 0.1) -> FusionPoseLoss:
    return FusionPoseLoss(heatmap_weight=heatmap_weight, offset_weight=offset_weight, variance_weight=variance_weight,
                          target_sigma=target_sigma)
