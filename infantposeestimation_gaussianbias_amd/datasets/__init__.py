"""Datasets module: the T1 target generator, the device input pipeline (affine crop / flip / normalise as one kernel per batch,
datasets/transforms.py) and a COCO loader that needs neither pycocotools nor OpenCV (JSON + PIL, datasets/coco_dataset.py)."""
import os

from .generate_heatmap import HeatmapGenerator, generate_dense_target, generate_target
from .synthetic import SyntheticLoader, synthetic_batch


def build_dataloader(cfg, is_train: bool = True):
    """Reference signature (datasets/coco_dataset.py:253).  POSE_SYNTHETIC=1: device-resident synthetic batches (same batch-dict keys,
    no files needed).  Otherwise the COCO annotation file must exist (FileNotFoundError like the reference); images are decoded on
    DataLoader workers and cropped / normalised / turned into targets on the device."""
    ann = os.path.join(cfg.data.data_root, cfg.data.train_ann if is_train else cfg.data.val_ann)
    if os.environ.get("POSE_SYNTHETIC", "0") == "1":
        return SyntheticLoader(cfg, n_batches=int(os.environ.get("POSE_SYNTHETIC_BATCHES", "10")))
    if not os.path.isfile(ann):
        raise FileNotFoundError(f"{ann}: annotation file not found (set POSE_SYNTHETIC=1 for device-resident synthetic batches)")
    from .coco_dataset import build_coco_dataloader
    return build_coco_dataloader(cfg, is_train)


from . import transforms  # noqa: E402,F401

__all__ = ["build_dataloader", "generate_target", "generate_dense_target", "HeatmapGenerator", "SyntheticLoader", "synthetic_batch", "transforms"]
