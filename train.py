#!/usr/bin/env python3
"""Training script (drop-in for the reference's train.py: same flags, same checkpoint dict).

    python train.py --data_root data/coco/ --batch_size 64 --epochs 210 --lr 5e-4 [--resume ckpt.pth] [--config hrformer_small]
    torchrun --nproc-per-node 8 train.py ...          # data parallel, one process per GPU (RCCL)

Differences from the reference loop (train.py:131-228): no per-step `loss.item()` host sync (losses are read back only at the
log interval), bf16 instead of fp16+GradScaler, fused AdamW over a flat parameter buffer, optional hipGraph replay
(`--graph`).  With POSE_SYNTHETIC=1 the loader yields device-resident synthetic batches (no COCO on disk needed).
"""
import argparse
import logging
import os
import sys
import time
from datetime import datetime, timedelta

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from infantposeestimation_gaussianbias_amd import engine  # noqa: E402
from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.datasets import build_dataloader  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402
from infantposeestimation_gaussianbias_amd.utils import AverageMeter  # noqa: E402


def set_seed(seed: int):
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def save_checkpoint(model, trainer, epoch, metrics, output_dir, is_best=False):
    """Same dict as the reference (train.py:351-357); optimizer state is exported in torch.optim.AdamW's format."""
    os.makedirs(output_dir, exist_ok=True)
    ckpt = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": trainer.opt.state_dict(),
            "scheduler_state_dict": trainer.sched.state_dict(), "metrics": metrics}
    torch.save(ckpt, os.path.join(output_dir, "latest.pth"))
    if is_best:
        torch.save(ckpt, os.path.join(output_dir, "best.pth"))
    if epoch % 10 == 0:
        torch.save(ckpt, os.path.join(output_dir, f"epoch_{epoch}.pth"))


def train_one_epoch(trainer, loader, epoch, cfg, logger, rank):
    loss_meter, batch_time = AverageMeter("Loss", ":.4f"), AverageMeter("Time", ":.3f")
    n = len(loader)
    log_interval = max(1, n // 10)
    end = time.time()
    for i, batch in enumerate(loader):
        dev = next(trainer.model.parameters()).device
        batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        out = trainer.step(batch)
        if i % log_interval == 0 or i == n - 1:          # the only host syncs of the epoch
            loss_meter.update(float(out["loss"].detach()), batch["img"].size(0))
            batch_time.update((time.time() - end) / (log_interval if i else 1))
            end = time.time()
            if rank == 0:
                msg = (f"Epoch [{epoch}][{i}/{n}] Loss: {loss_meter.val:.4f} ({loss_meter.avg:.4f}) "
                       f"LR: {trainer.opt.lr:.6f} Time: {batch_time.val:.3f}s")
                if "losses" in out:
                    msg += " | " + ", ".join(f"{k}: {float(v.detach()):.4f}" for k, v in out["losses"].items() if k != "total_loss")
                logger.info(msg)
    return loss_meter.avg


def main(args):
    cfg = get_config(args.config) if args.config else get_config()
    if args.data_root:
        cfg.data.data_root = args.data_root
    if args.batch_size:
        cfg.train.batch_size = args.batch_size
    if args.epochs:
        cfg.train.max_epochs = args.epochs
    if args.lr:
        cfg.train.lr = args.lr
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    logger = logging.getLogger("train")
    set_seed(cfg.seed + rank)
    out_dir = os.path.join(cfg.train.checkpoint_dir, f"{cfg.exp_name}_{datetime.now().strftime('%Y%m%d_%H%M%S')}")
    loader = build_dataloader(cfg, is_train=True)
    model = build_model(cfg).to(torch.device("cuda", local))
    trainer = engine.Trainer(model, cfg, iters_per_epoch=len(loader), use_graph=args.graph, graph_streams=args.graph)
    start_epoch, best_ap = 0, 0.0
    if args.resume and os.path.isfile(args.resume):
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=True)
        model.load_state_dict(ckpt["model_state_dict"])
        trainer.opt.load_state_dict(ckpt["optimizer_state_dict"])
        trainer.sched.load_state_dict(ckpt["scheduler_state_dict"])
        start_epoch, best_ap = ckpt["epoch"] + 1, ckpt.get("metrics", {}).get("AP", 0.0)
        logger.info(f"Resumed from epoch {start_epoch}")
    val_loader = None
    for epoch in range(start_epoch, cfg.train.max_epochs):
        loss = train_one_epoch(trainer, loader, epoch, cfg, logger, rank)
        if rank == 0 and ((epoch + 1) % cfg.train.val_interval == 0 or epoch == cfg.train.max_epochs - 1):
            # validation inside the training loop (train.py:231-325, 437-452 of the reference): AP decides `best.pth`
            from validate import validate
            val_loader = val_loader or build_dataloader(cfg, is_train=False)
            metrics, _ = validate(model, val_loader, torch.device("cuda", local), cfg, logger, flip_test=False)
            model.train()
            is_best = metrics["AP"] > best_ap
            best_ap = max(best_ap, metrics["AP"])
            save_checkpoint(model, trainer, epoch, dict({k: float(v) for k, v in metrics.items()}, train_loss=float(loss)), out_dir, is_best=is_best)
        if world > 1 and ((epoch + 1) % cfg.train.val_interval == 0 or epoch == cfg.train.max_epochs - 1):
            # The other ranks wait for rank 0's validation on the HOST (a key in the rendezvous store), not inside a collective: a
            # dist.barrier() or the next epoch's first all-reduce would sit enqueued for the whole validation and trip the RCCL
            # watchdog timeout.  Unverified on hardware until a >= 2-GPU run exists (SCALE has been skipped so far).
            store = dist.distributed_c10d._get_default_store()
            key = f"pose/validated/{epoch}"
            if rank == 0:
                store.set(key, "1")
            else:
                store.wait([key], timedelta(hours=12))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Train Pose Estimation Model")
    p.add_argument("--data_root", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--config", type=str, default=None, help="preset name or legacy yaml (extension over the reference)")
    p.add_argument("--graph", action="store_true", help="replay the step as a captured hipGraph")
    main(p.parse_args())
