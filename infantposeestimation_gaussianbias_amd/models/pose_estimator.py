"""PoseEstimator / build_model: the drop-in surface of the reference's models/pose_estimator.py."""
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import hipops
from .. import dispatch as nnops
from .fusion_head import FusionPoseLoss, HeatmapRegressionHead, SoftArgmax2D
from .hrformer import hrformer_base, hrformer_small
from .hrnet import hrnet_w18, hrnet_w32, hrnet_w48

_BACKBONES = {"hrnet_w32": (hrnet_w32, 32), "hrnet_w48": (hrnet_w48, 48), "hrnet_w18": (hrnet_w18, 18),
              "hrformer_base": (hrformer_base, 78), "hrformer_small": (hrformer_small, 32)}


class HeatmapHead(nn.Module):
    """Optional SimpleBaseline-style deconv stack (ConvTranspose2d stride 2 + BN + ReLU per layer), then a 1x1 conv to K maps; init
    N(0, 0.001) / BN 1, 0 (pose_estimator.py:22-99).  The reference's builders use num_deconv_layers = 0; the stack runs as stacked
    parity-class convolutions + pixel shuffle (nnops.deconv_bn_relu).  Kernel 3 raises, as in the reference (output_padding -1)."""

    def __init__(self, in_channels: int, out_channels: int, num_deconv_layers: int = 0, num_deconv_filters=(256, 256, 256),
                 num_deconv_kernels=(4, 4, 4)):
        super().__init__()
        self.in_channels, self.out_channels, self.deconv = in_channels, out_channels, None
        final_in = in_channels
        if num_deconv_layers > 0:
            layers = []
            for i in range(num_deconv_layers):
                cin, cout, k = (in_channels if i == 0 else num_deconv_filters[i - 1]), num_deconv_filters[i], num_deconv_kernels[i]
                pad = (k - 1) // 2
                if k - 2 * pad - 2 < 0:      # kernel 3: the reference builds the layer and fails in its first forward ("negative output_padding")
                    raise ValueError(f"HeatmapHead: deconv kernel {k} gives output_padding {k - 2 * pad - 2} by the reference's rule")
                layers += [nn.ConvTranspose2d(cin, cout, k, stride=2, padding=pad, output_padding=k - 2 * pad - 2, bias=False),
                           nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]       # parameter holders with the reference's Sequential indices
            self.deconv = nn.Sequential(*layers)
            final_in = num_deconv_filters[num_deconv_layers - 1]
        self.final_layer = nn.Conv2d(final_in, out_channels, 1)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        with nnops.scope(self):
            t = nnops.from_public(x)
            if self.deconv is not None:
                for i in range(0, len(self.deconv), 3):
                    t = nnops.deconv_bn_relu(t, self.deconv[i], self.deconv[i + 1], self.training)
            return nnops.head_out(t, self.final_layer)


class KeypointMSELoss(nn.Module):
    def __init__(self, use_target_weight: bool = True):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, pred, target, target_weight=None):
        w = target_weight.float() if (self.use_target_weight and target_weight is not None) else None
        return hipops.pixel_loss(pred.float(), target.float(), w, 0)


class PoseEstimator(nn.Module):
    def __init__(self, backbone: str = "hrformer_base", num_keypoints: int = 17, pretrained: bool = True,
                 head_type: str = "fusion", use_fusion_loss: bool = True):
        super().__init__()
        if backbone not in _BACKBONES:
            raise ValueError(f"Unknown backbone: {backbone}")
        ctor, in_channels = _BACKBONES[backbone]
        self.head_type = head_type
        self.use_fusion_loss = use_fusion_loss and head_type == "fusion"
        self.backbone = ctor(pretrained=pretrained)
        if head_type == "fusion":
            self.head = HeatmapRegressionHead(in_channels, num_keypoints, 256, True)
            self.loss_fn = FusionPoseLoss(1.0, 1.0, 0.5, 0.1, 0.05, 0.05, 2.0, True)
        else:
            self.head = HeatmapHead(in_channels, num_keypoints, 0)
            self.loss_fn = KeypointMSELoss(True)
        self.num_keypoints = num_keypoints
        self.soft_argmax = SoftArgmax2D()

    @classmethod
    def from_backbone(cls, backbone_module: nn.Module, in_channels: int, num_keypoints: int, head_type: str, use_fusion_loss: bool):
        """Same composition as __init__ around an already-built backbone (used for 8-aligned padded twins)."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        self.head_type = head_type
        self.use_fusion_loss = use_fusion_loss and head_type == "fusion"
        self.backbone = backbone_module
        if head_type == "fusion":
            self.head = HeatmapRegressionHead(in_channels, num_keypoints, 256, True)
            self.loss_fn = FusionPoseLoss(1.0, 1.0, 0.5, 0.1, 0.05, 0.05, 2.0, True)
        else:
            self.head = HeatmapHead(in_channels, num_keypoints, 0)
            self.loss_fn = KeypointMSELoss(True)
        self.num_keypoints = num_keypoints
        self.soft_argmax = SoftArgmax2D()
        return self

    def forward(self, x, target=None, target_weight=None, gt_keypoints=None, input_size: Tuple[int, int] = (192, 256)) -> Dict[str, torch.Tensor]:
        tw = nnops.padded_twin(self)
        if tw is not None:                       # backbone channels not multiples of 8: the whole estimator runs as its 8-aligned twin
            return tw.run(x, target, target_weight, gt_keypoints, input_size)
        with nnops.scope(self):
            feats = self.backbone(x)
            output = dict(self.head(feats)) if self.head_type == "fusion" else {"heatmaps": self.head(feats)}
        if target is not None:
            if self.head_type == "fusion" and self.use_fusion_loss and gt_keypoints is not None:
                H, W = output["heatmaps"].shape[2:]
                losses = self.loss_fn(output, target, target_weight, gt_keypoints, input_size, (H, W))
                output["loss"], output["losses"] = losses["total_loss"], losses
            elif self.head_type == "fusion":
                raise ValueError("fusion head without gt_keypoints: the reference would call FusionPoseLoss with 3 arguments and fail")
            else:
                output["loss"] = self.loss_fn(output["heatmaps"], target, target_weight)
        return output

    @torch.no_grad()
    def inference(self, x, flip: bool = True, flip_pairs: Optional[list] = None):
        """-> keypoints (B,K,2) heat-px, scores (B,K); flip test as pose_estimator.py:275-329 (offsets of the un-flipped pass)."""
        if flip and flip_pairs is not None and not self.training and os.environ.get("POSE_FLIP_BATCHED", "1") != "0":
            # eval mode: every sample is independent of the rest of its batch (running BatchNorm statistics, per-sample windows), so the two
            # passes of the flip test are ONE forward over [x ; flip(x)] -- same numbers per sample, half the launches, twice the rows per launch
            B = x.shape[0]
            both = self.forward(torch.cat([x, torch.flip(x, dims=[-1])], 0))
            out = {k: (v[:B] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 * B else v) for k, v in both.items()}
            hm = hipops.flip_merge(out["heatmaps"], both["heatmaps"][B:], self._flip_partner(flip_pairs, x.device))
        else:
            out = self.forward(x)
            hm = out["heatmaps"]
            if flip and flip_pairs is not None:
                hm_f = self.forward(torch.flip(x, dims=[-1]))["heatmaps"]
                hm = hipops.flip_merge(hm, hm_f, self._flip_partner(flip_pairs, hm.device))
        if self.head_type == "fusion":
            o = dict(out)
            o["heatmaps"] = hm
            return self.head.decode(o, apply_offset=True)
        return self.decode_heatmaps(hm)

    def _flip_partner(self, flip_pairs, device):
        """int32 (K,) left/right partner of every keypoint, cached on the device (no host->device copy per call: inference is
        hipGraph-capturable)."""
        key = (tuple(tuple(p) for p in flip_pairs), str(device))
        cache = self.__dict__.setdefault("_partner_cache", {})
        if key not in cache:
            partner = torch.arange(self.num_keypoints, dtype=torch.int32)
            for a, b in flip_pairs:
                partner[a], partner[b] = b, a
            cache[key] = partner.to(device)
        return cache[key]

    @staticmethod
    @torch.no_grad()
    def decode_heatmaps(heatmaps, shift: bool = True):
        _, mv, co = hipops.argmax_decode(heatmaps.float(), 1 if shift else 0)
        return co, mv


def build_model(cfg) -> PoseEstimator:
    return PoseEstimator(backbone=cfg.model.backbone, num_keypoints=cfg.model.num_keypoints, pretrained=cfg.model.pretrained,
                         head_type=getattr(cfg.model, "head_type", "fusion"), use_fusion_loss=getattr(cfg.model, "use_fusion_loss", True))
