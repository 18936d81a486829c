#!/usr/bin/env python3
"""Stand-alone launches of the head convolution (3x3 256->256 @64x48, B=64) for rocprofv3 --pmc / --kernel-trace runs:
forward with and without the BatchNorm-statistics epilogue, data gradient, weight gradient; ITER launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import nnops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
ITER = int(os.environ.get("ITER", "5"))
B, H, W, C = int(os.environ.get("PB", "64")), 64, 48, 256


class Holder(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c


m = Holder(torch.nn.Conv2d(C, C, 3, 1, 1, bias=False)).to(DEV)
x = torch.randn(B, H, W, C, device=DEV).to(BF)
g = torch.randn(B, H, W, C, device=DEV).to(BF)
with nnops.use_weights(m) as wc:
    wf, wd = wc.fwd[id(m.c.weight)], wc.dgrad[id(m.c.weight)]
    for what in sys.argv[1:] or ["fwd", "fwd_nostats", "dgrad"]:
        for _ in range(ITER):
            if what == "fwd":
                nnops._conv_raw(x, wf, C, 3, 1, True)
            elif what == "fwd_nostats":
                nnops._conv_raw(x, wf, C, 3, 1, False)
            elif what == "dgrad":
                nnops._conv_dgrad(g, wd, C, 3, 1, (H, W))
            elif what == "c64":          # 3x3 64 -> 64 @64x48 forward + statistics (k_conv3h, or k_igemm2 with PK_CONV3H=0)
                if "x64" not in globals():
                    x64 = torch.randn(B, H, W, 64, device=DEV).to(BF)
                    w64 = torch.randn(64, 9, 64, device=DEV).to(BF)
                nnops._conv_raw(x64, w64, 64, 3, 1, True)
            elif what == "wgrad":
                nnops._wgrad(x, g, C, C, 3, 1, (B, H, W, H, W))
        torch.cuda.synchronize()
print("done")
