#!/usr/bin/env python3
"""Isolated timing of the N <= 32 deep convs (chunk-major K order): 3x3 256->32 / 128->32 at the HRFormer-small shapes.  python scripts/bench_n32_conv.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import nnops  # noqa: E402
from bench import time_kernel  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
for (B, H, W, Cin, Cout) in [(64, 64, 48, 256, 32), (64, 32, 24, 128, 32), (64, 64, 48, 128, 32)]:
    x, w = torch.randn(B, H, W, Cin, device=DEV).to(BF), torch.randn(Cout, 9, Cin, device=DEV).to(BF)
    f = time_kernel(lambda: nnops._conv_raw(x, w, Cout, 3, 1, True))
    print(f"conv3x3 {Cin}->{Cout} @{H}x{W} B={B}: fwd+stats {f * 1e6:7.1f} us", flush=True)
