#!/usr/bin/env python3
"""Isolated timing (rocprof-free, hipGraph of 20 launches) of the low-resolution deep-K convs of HRNet-W32 (cfg 4, B = 32): forward with
statistics and data gradient.  python scripts/bench_lowres_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import nnops  # noqa: E402
from bench import time_kernel  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
B = int(os.environ.get("PB", "32"))
for (H, W, C) in [(12, 9, 256), (24, 18, 128), (48, 36, 64), (16, 12, 128), (8, 6, 256)]:
    x, w = torch.randn(B, H, W, C, device=DEV).to(BF), torch.randn(C, 9, C, device=DEV).to(BF)
    f = time_kernel(lambda: nnops._conv_raw(x, w, C, 3, 1, True))
    d = time_kernel(lambda: nnops._conv_dgrad(x, w, C, 3, 1, (H, W)))
    fl = 2.0 * B * H * W * C * 9 * C
    print(f"conv3x3 {C}->{C} @{H}x{W} B={B} (M={B * H * W}, {9 * C // 64} K-steps): fwd+stats {f * 1e6:7.1f} us ({fl / f / 1e12:6.1f} TFLOP/s)  dgrad {d * 1e6:7.1f} us", flush=True)
