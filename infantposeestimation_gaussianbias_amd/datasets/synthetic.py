"""Synthetic batches in the reference's batch-dict format (datasets/coco_dataset.py:169-183), resident on device.

SURVEY §8d: img ~ N(0,1) (B,3,H,W); keypoints ~ U([0,W)x[0,H)); vis in {0,1,2} with p=(.15,.25,.60);
target/target_weight come from the T1 kernel.  Seed = 1234 + rank.
"""
import torch

from .generate_heatmap import generate_target


def synthetic_batch(batch_size, input_size=(192, 256), heatmap_size=(48, 64), num_keypoints=17, sigma=2.0, device="cuda", seed=1234,
                    id_base=0):
    """`id_base`: first image / annotation id of the batch (a loader numbers its batches consecutively, so that an evaluator keyed by
    image id never merges instances of different batches)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    W, H = input_size
    img = torch.randn(batch_size, 3, H, W, generator=g)
    kp = torch.rand(batch_size, num_keypoints, 2, generator=g) * torch.tensor([float(W), float(H)])
    vis = torch.multinomial(torch.tensor([0.15, 0.25, 0.60]), batch_size * num_keypoints, replacement=True, generator=g)
    vis = vis.reshape(batch_size, num_keypoints).float()
    img, kp, vis = img.to(device), kp.to(device), vis.to(device)
    target, weight = generate_target(kp, vis, input_size, heatmap_size, sigma)
    return {"img": img, "target": target, "target_weight": weight, "keypoints": kp, "keypoints_visible": vis,
            "meta": {"image_id": torch.arange(batch_size) + id_base, "ann_id": torch.arange(batch_size) + id_base,
                     "center": torch.tensor([[W / 2.0, H / 2.0]]).repeat(batch_size, 1),
                     "scale": torch.tensor([[float(W), float(H)]]).repeat(batch_size, 1),
                     "bbox": torch.tensor([[0.0, 0.0, float(W), float(H)]]).repeat(batch_size, 1),
                     "area": torch.full((batch_size,), float(W * H))}}


class SyntheticLoader:
    """Iterable of `n_batches` device-resident synthetic batches (stands in for build_dataloader when no COCO is on disk)."""

    def __init__(self, cfg, n_batches=10, device="cuda", seed=1234, rank=None):
        import os
        rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        # every rank draws its own shard: batch i of rank r is seeded seed + r * n_batches + i
        self.cfg, self.n, self.device, self.seed = cfg, n_batches, device, seed + rank * n_batches
        self.dataset = range(n_batches * cfg.train.batch_size)

    def __len__(self):
        return self.n

    def __iter__(self):
        d = self.cfg.data
        for i in range(self.n):
            yield synthetic_batch(self.cfg.train.batch_size, d.input_size, d.heatmap_size, d.num_keypoints, d.sigma, self.device, self.seed + i,
                                  id_base=i * self.cfg.train.batch_size)
