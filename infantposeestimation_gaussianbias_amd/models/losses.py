"""Named loss classes of the reference's models/losses.py, on libposekernels reductions."""
import torch
import torch.nn as nn

from .. import hipops


class FusedPoseLoss(nn.Module):
    def __init__(self, use_target_weight=True, loss_type="mse"):
        super().__init__()
        if loss_type not in ("mse", "smoothl1"):
            raise ValueError(f"Unsupported loss type: {loss_type}")
        self.use_target_weight, self.loss_type = use_target_weight, loss_type

    def forward(self, pred_heatmaps, target_heatmaps, target_weight=None):
        w = target_weight.float() if (self.use_target_weight and target_weight is not None) else None
        return hipops.pixel_loss(pred_heatmaps.float(), target_heatmaps.float(), w, 1 if self.loss_type == "mse" else 2)


class MorphologyShapeLoss(nn.Module):
    """lambda_var*MSE(var) + lambda_mean*MSE(mean) of hm/(sum+1e-8), weighted, mean (losses.py:50-135).
    Statistics come from one reduction kernel per tensor (pk_spatial_stats) with a matching backward kernel; the few-hundred
    element combination of the statistics is torch elementwise on the device."""

    def __init__(self, lambda_variance=1.0, lambda_mean=0.5):
        super().__init__()
        self.lambda_variance, self.lambda_mean = lambda_variance, lambda_mean

    def compute_spatial_statistics(self, heatmaps):
        return hipops.spatial_stats(heatmaps.float())

    def forward(self, pred_heatmaps, target_heatmaps, target_weight=None):
        pm, pv = self.compute_spatial_statistics(pred_heatmaps)
        with torch.no_grad():
            tm, tv = self.compute_spatial_statistics(target_heatmaps)
        e = self.lambda_variance * (pv - tv) ** 2 + self.lambda_mean * (pm - tm) ** 2
        if target_weight is not None:
            e = e * target_weight.view(e.shape[0], e.shape[1], 1)
        return e.mean()


class OffsetRegressionLoss(nn.Module):
    def __init__(self, loss_type="smoothl1"):
        super().__init__()
        if loss_type not in ("smoothl1", "l1", "mse"):
            raise ValueError(f"Unsupported loss type: {loss_type}")
        self.loss_type = loss_type

    def forward(self, pred_coords, target_coords, target_weight=None):
        # (B,K,2) vectors: a few hundred floats -> torch elementwise on device (not a hot-path reduction)
        d = pred_coords - target_coords
        e = {"smoothl1": torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5), "l1": d.abs(), "mse": d * d}[self.loss_type]
        if target_weight is not None:
            e = e * target_weight.view(e.shape[0], e.shape[1], 1)
        return e.mean()


class JointsMSELoss(nn.Module):
    def __init__(self, use_target_weight=True):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight):
        return hipops.pixel_loss(output.float(), target.float(), target_weight.float() if self.use_target_weight else None, 3)


class CombinedLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.heatmap_loss = FusedPoseLoss(True, "mse")
        self.morph_loss = MorphologyShapeLoss(config.LOSS.MORPH_LAMBDA, 0.5)
        self.regression_loss = OffsetRegressionLoss("smoothl1")
        self.w_heatmap, self.w_morph, self.w_reg = 1.0, config.LOSS.MORPH_WEIGHT, config.LOSS.REG_WEIGHT

    def forward(self, predictions, targets):
        losses, w = {}, targets.get("weights")
        if "heatmaps" in predictions and "heatmaps" in targets:
            losses["heatmap"] = self.heatmap_loss(predictions["heatmaps"], targets["heatmaps"], w)
            losses["morph"] = self.morph_loss(predictions["heatmaps"], targets["heatmaps"], w)
        if "coords" in predictions and "coords" in targets:
            losses["regression"] = self.regression_loss(predictions["coords"], targets["coords"], w)
        if "refined_coords" in predictions and "coords" in targets:
            losses["refined"] = self.regression_loss(predictions["refined_coords"], targets["coords"], w)
        total = (self.w_heatmap * losses.get("heatmap", 0) + self.w_morph * losses.get("morph", 0) +
                 self.w_reg * losses.get("regression", 0) + self.w_reg * losses.get("refined", 0))
        losses["total"] = total
        return total, losses


def build_loss(config):
    return CombinedLoss(config)
