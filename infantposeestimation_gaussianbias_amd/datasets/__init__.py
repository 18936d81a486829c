"""Datasets module. The hot-path piece is the target generator; COCO file I/O + cv2 augmentation stay host-side
work the reference does with pycocotools/OpenCV (out of this round's scope)."""
import os

from .generate_heatmap import HeatmapGenerator, generate_dense_target, generate_target
from .synthetic import SyntheticLoader, synthetic_batch


def build_dataloader(cfg, is_train: bool = True):
    """Reference signature (datasets/coco_dataset.py:253). Without COCO + pycocotools + cv2 on the machine this
    returns the synthetic loader (same batch-dict keys) when POSE_SYNTHETIC=1, else raises like the reference would."""
    ann = os.path.join(cfg.data.data_root, cfg.data.train_ann if is_train else cfg.data.val_ann)
    if os.environ.get("POSE_SYNTHETIC", "0") == "1":
        return SyntheticLoader(cfg, n_batches=int(os.environ.get("POSE_SYNTHETIC_BATCHES", "10")))
    raise FileNotFoundError(f"{ann}: COCO data loading (pycocotools + OpenCV warps) is not part of this build yet; "
                            "set POSE_SYNTHETIC=1 for device-resident synthetic batches")


__all__ = ["build_dataloader", "generate_target", "generate_dense_target", "HeatmapGenerator", "SyntheticLoader", "synthetic_batch"]
