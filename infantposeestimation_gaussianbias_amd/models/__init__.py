"""Models module (same exports as the reference's models/__init__.py, plus hrnet_w18)."""
from .hrnet import HRNet, hrnet_w18, hrnet_w32, hrnet_w48
from .hrformer import HRFormer, hrformer_base, hrformer_small
from .fusion_head import (HeatmapRegressionHead, FusionPoseLoss, GaussianDistributionConstraint, SoftArgmax2D,
                          SubPixelRefinement, build_fusion_head, build_fusion_loss)
from .pose_estimator import HeatmapHead, KeypointMSELoss, PoseEstimator, build_model

__all__ = ['HRNet', 'hrnet_w32', 'hrnet_w48', 'hrnet_w18', 'HRFormer', 'hrformer_base', 'hrformer_small',
           'HeatmapRegressionHead', 'FusionPoseLoss', 'GaussianDistributionConstraint', 'SoftArgmax2D',
           'SubPixelRefinement', 'build_fusion_head', 'build_fusion_loss', 'HeatmapHead', 'KeypointMSELoss',
           'PoseEstimator', 'build_model']
