"""MI355X-native top-down pose-estimation hot path (HRFormer / HRNet), drop-in at the reference's Python surface.

    from infantposeestimation_gaussianbias_amd.configs import get_config
    from infantposeestimation_gaussianbias_amd.models import build_model, PoseEstimator

Importing the package loads libposekernels.so (hand-written HIP for gfx950) and fails loudly if it is not built.
"""
from . import _lib  # noqa: F401  (raises PoseKernelError when the HIP library is missing)
from . import configs, datasets, models, utils  # noqa: F401

__version__ = "0.1.0"
