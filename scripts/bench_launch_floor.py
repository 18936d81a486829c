#!/usr/bin/env python3
"""Per-launch floor inside a hipGraph: 20 dependent launches of (a) an empty kernel (pk_marker, 1 workgroup), (b) the smallest useful
kernels of the step, timed as bench.time_kernel does (graph replay, HIP events).  python scripts/bench_launch_floor.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import nnops  # noqa: E402
from infantposeestimation_gaussianbias_amd._lib import call, stream_ptr  # noqa: E402
from bench import time_kernel  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
print(f"empty kernel (1 workgroup):           {time_kernel(lambda: call('pk_marker', 1, stream_ptr())) * 1e6:6.2f} us per launch")
print(f"empty kernel (1024 workgroups):       {time_kernel(lambda: call('pk_marker', 1024, stream_ptr())) * 1e6:6.2f} us per launch")
x = torch.randn(32, 12, 9, 256, device=DEV).to(BF)
y = torch.empty_like(x)
sc, sh = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
print(f"k_bn_act 3 456 x 256 (1.8 MB):        {time_kernel(lambda: call('pk_bn_act', x, sc, sh, None, y, 3456, 256, 1, stream_ptr())) * 1e6:6.2f} us per launch")
w = torch.randn(256, 1, 256, device=DEV).to(BF)
print(f"conv 1x1 256->256 @12x9 (4 K-steps):  {time_kernel(lambda: nnops._conv_raw(x, w, 256, 1, 1, False)) * 1e6:6.2f} us per launch")
w3 = torch.randn(256, 9, 256, device=DEV).to(BF)
print(f"conv 3x3 256->256 @12x9 (36 K-steps): {time_kernel(lambda: nnops._conv_raw(x, w3, 256, 3, 1, False)) * 1e6:6.2f} us per launch")
