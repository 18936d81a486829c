#!/usr/bin/env python3
"""Flip-test inference throughput of HRFormer-base + fusion head, K=13, 384x288 (BASELINE config 5) on one MI355X.
Not the headline metric (bench.py measures that); this exercises the 8-aligned padded twin at full size.

    python scripts/bench_infer_base.py [batch] [iters]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import dispatch  # noqa: E402
from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    cfg = get_config("preemie")
    torch.manual_seed(0)
    model = build_model(cfg).cuda().eval()
    x = torch.randn(B, 3, cfg.data.input_size[1], cfg.data.input_size[0], device="cuda")
    pairs = [(1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12)]
    with torch.no_grad():
        for _ in range(3):
            kp, sc = model.inference(x, flip=True, flip_pairs=pairs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            kp, sc = model.inference(x, flip=True, flip_pairs=pairs)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{dispatch.backend_name(model)}: HRFormer-base K={cfg.model.num_keypoints} {cfg.data.input_size} flip-test inference, B={B}: "
          f"{dt * 1e3:.1f} ms/batch = {B / dt:.1f} img/s; keypoints {tuple(kp.shape)}, finite={bool(torch.isfinite(kp).all())}, "
          f"HBM reserved {torch.cuda.max_memory_reserved() / 2 ** 30:.1f} GB")


if __name__ == "__main__":
    main()
