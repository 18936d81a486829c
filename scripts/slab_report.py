#!/usr/bin/env python3
"""Which deferred slab reductions make up k_reduce_many's traffic in one BASELINE cfg-2 step: rows of the reduction table by bytes read."""
import os
import sys
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import engine, nnops  # noqa: E402
from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402

cfg = get_config("hrformer_small")
cfg.train.batch_size = 64
model = build_model(cfg).to("cuda")
batch = synthetic_batch(64, cfg.data.input_size, cfg.data.heatmap_size, 17, cfg.data.sigma, "cuda", seed=1)
tr = engine.Trainer(model, cfg, iters_per_epoch=1000, use_graph=False, graph_streams=True)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
key = max(nnops._TABLES, key=len)
tot = sum(r[3] * r[4] * 4 for r in key)
print(f"{len(key)} reductions, {tot / 1e6:.0f} MB of slabs per step")
by = collections.Counter()
for r in key:
    S, K = r[3], r[4]
    by[(S, K)] += S * K * 4
for (S, K), b in by.most_common(25):
    n = sum(1 for r in key if (r[3], r[4]) == (S, K))
    print(f"  {b / 1e6:8.1f} MB  {n:3d} x  S={S:5d} slabs of K={K:7d} outputs")
