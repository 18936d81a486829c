#!/usr/bin/env python3
"""Stand-alone launches for the rocprofv3 --pmc traffic passes (FETCH_SIZE / WRITE_SIZE, one counter per pass).

    python3 scripts/prof_kernels.py isolated    the roofline table's GEMM-shaped probes, each kernel name on ONE shape, 3 launches each
    python3 scripts/prof_kernels.py wide        the forward-only wide fused halves (k_attn_fwd_w, k_mlp_fwd_w) at their cfg-5 shapes
    python3 scripts/prof_kernels.py step        two eager training steps of BASELINE cfg 2 (B = 64): every kernel of the step at its real
                                                shapes (per-kernel-name means are exact for single-shape kernels: k_reduce_many, k_conv8p,
                                                k_wgrad3, the C = 32 / 64 fused block kernels, k_adamw)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import engine, nnops  # noqa: E402
from infantposeestimation_gaussianbias_amd._lib import call, lib, stream_ptr  # noqa: E402
from infantposeestimation_gaussianbias_amd.configs import get_config  # noqa: E402
from infantposeestimation_gaussianbias_amd.datasets import synthetic_batch  # noqa: E402
from infantposeestimation_gaussianbias_amd.models import build_model  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
mode = sys.argv[1] if len(sys.argv) > 1 else "isolated"
B, H, W = 64, 64, 48
M = B * H * W
torch.manual_seed(0)
if mode == "wide":
    # the forward-only wide fused halves at their BASELINE cfg 5 shapes (HRFormer-base twin, 2 x 32 samples: 96 x 72 tokens of C = 80, 48 x 36 of C = 160)
    from infantposeestimation_gaussianbias_amd.models.hrformer import HRFormerBlock
    for C, heads, Hh, Ww in ((80, 2, 96, 72), (160, 4, 48, 36)):
        blk = HRFormerBlock(C, heads).to(DEV)
        blk.c_real, blk.attn_scale = C - 2, float(C // heads - 1) ** -0.5
        x = (torch.randn(64, Hh, Ww, C, device=DEV) * 1.5).to(BF)          # the flip test batches x and flip(x): 2 x 32 samples
        a, m = blk.attn, blk.mlp
        with torch.no_grad(), nnops.use_weights(blk):
            for _ in range(3):
                if nnops.wide_attn_enabled(C, heads):
                    nnops.attn_half_wide_forward(x, blk.norm1.weight, blk.norm1.bias, nnops.rel_table(a, heads), a.qkv.weight, a.qkv.bias, a.proj.weight,
                                                 a.proj.bias, None, heads, blk.c_real, blk.attn_scale)
                nnops.mlp_half_wide_forward(x, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, None, blk.c_real)
                torch.cuda.synchronize()
elif mode == "step":
    cfg = get_config("hrformer_small")
    cfg.train.batch_size = B
    model = build_model(cfg).to(DEV)
    batch = synthetic_batch(B, cfg.data.input_size, cfg.data.heatmap_size, 17, cfg.data.sigma, DEV, seed=1234)
    tr = engine.Trainer(model, cfg, iters_per_epoch=1000, use_graph=False, graph_streams=True)
    for _ in range(3):
        tr.step(batch)
    torch.cuda.synchronize()
else:
    rnd = lambda *s: torch.randn(*s, device=DEV).to(BF)
    x256, g256, x64 = rnd(B, H, W, 256), rnd(B, H, W, 256), rnd(B, H, W, 64)
    w256, wd256, w64, w32 = rnd(256, 9, 256), rnd(256, 9, 256), rnd(64, 9, 64), rnd(32, 9, 256)
    for _ in range(3):
        nnops._conv_raw(x256, w256, 256, 3, 1, True)                       # k_conv8p forward + statistics
        nnops._conv_dgrad(g256, wd256, 256, 3, 1, (H, W))                  # k_conv8p data gradient
        nnops._conv_raw(x64, w64, 64, 3, 1, True)                          # k_conv3h<64,64>
        nnops._conv_raw(x256, w32, 32, 3, 1, True)                         # k_igemm2<128,32,4,1,64>
        S = lib.pk_wgrad_slices(M, 256, 256, 3, 1, H, W, 0)
        ws = torch.empty(S * 256 * (9 * 256 + 1), device=DEV)
        call("pk_wgrad_bf16", x256, g256, ws, None, None, 0, None, None, None, 0, M, 256, 256, 3, 1, B, H, W, H, W, 0, stream_ptr())   # k_wgrad3 (slabs)
        torch.cuda.synchronize()
print("done", mode)
