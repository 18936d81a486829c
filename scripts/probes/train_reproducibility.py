#!/usr/bin/env python3
"""Two trainers from one seed, N captured steps each (branch streams, hipGraph replay): are losses and weights bit-identical?
PROBE_CFG = hrformer_small | hrnet_w32_384;  PROBE_N steps."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from infantposeestimation_gaussianbias_amd import dispatch, engine  # noqa: E402
from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402

name = os.environ.get("PROBE_CFG", "hrformer_small")
N = int(os.environ.get("PROBE_N", "40"))
cfg = get_config({"hrnet_w32_384": "hrnet_w32"}.get(name, name))
B = 64 if name == "hrformer_small" else 32
cfg.train.batch_size = B
batches = [synthetic_batch(B, cfg.data.input_size, cfg.data.heatmap_size, cfg.data.num_keypoints, cfg.data.sigma, "cuda", seed=100 + i) for i in range(3)]
runs = []
for rep in range(3):
    torch.manual_seed(7)
    model = build_model(cfg).to("cuda")
    tr = engine.Trainer(model, cfg, iters_per_epoch=1000, use_graph=True, graph_warmup=2, graph_streams=True)
    losses = [tr.step(batches[i % 3])["loss"].detach().clone() for i in range(N)]
    torch.cuda.synchronize()
    runs.append((torch.stack(losses).cpu(), tr.opt.flat.detach().cpu().clone()))
    del tr, model
dispatch.set_region_mode(False)
for k in (1, 2):
    same_l = torch.equal(runs[0][0], runs[k][0])
    same_w = torch.equal(runs[0][1], runs[k][1])
    first = next((i for i in range(N) if runs[0][0][i] != runs[k][0][i]), None)
    print(f"{name} B={B}, {N} steps: run {k} vs run 0: losses identical = {same_l} (first differing step {first}), weights identical = {same_w}, "
          f"max |dw| = {float((runs[0][1] - runs[k][1]).abs().max()):.3g}")
print("losses", [round(float(x), 4) for x in runs[0][0][:4]], "...", [round(float(x), 4) for x in runs[0][0][-2:]])
