#!/usr/bin/env python3
"""Diagnostic: does a kernel's result change when OTHER kernels run on the same chip at the same time (another stream)?
Each candidate is run alone (reference) and then repeatedly while a noise stream keeps the chip busy; outputs are compared bit for bit."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from infantposeestimation_gaussianbias_amd import nnops as N  # noqa: E402
from infantposeestimation_gaussianbias_amd.models.hrformer import HRFormerBlock  # noqa: E402

DEV, BF = torch.device("cuda:0"), torch.bfloat16
REPS = int(os.environ.get("PROBE_N", "30"))
torch.manual_seed(0)


def block(C, heads):
    b = HRFormerBlock(C, heads)
    with torch.no_grad():
        for p in b.parameters():
            p.copy_((p * 4 if p.dim() > 1 else p + 0.1 * torch.randn_like(p)).to(BF).float())
        b.attn.relative_position_bias_table.copy_((torch.randn(169, heads) * 0.5).to(BF).float())
    return b.to(DEV).eval()


def attn_args(b, heads):
    a = b.attn
    return (b.norm1.weight, b.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, None, heads)


def mlp_args(b):
    m = b.mlp
    return (b.norm2.weight, b.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, None)


b80, b160, b32, b64 = block(80, 2), block(160, 4), block(32, 1), block(64, 2)
x80 = (torch.randn(16, 96, 72, 80) * 1.5 + 0.2).to(DEV, BF)
x160 = (torch.randn(16, 48, 36, 160) * 1.5 + 0.2).to(DEV, BF)
x32 = (torch.randn(16, 64, 48, 32) * 1.5).to(DEV, BF)
x64 = (torch.randn(16, 32, 24, 64) * 1.5).to(DEV, BF)
big = torch.randn(4096, 4096, device=DEV, dtype=BF)
xn = (torch.randn(16, 48, 36, 160) * 1.5).to(DEV, BF)

cands = {
    "k_attn_fwd_w (C=80)": lambda: (b80, lambda: N.attn_half_wide_forward(x80, *attn_args(b80, 2), 78, 39.0 ** -0.5)),
    "k_mlp_fwd_w (C=80)": lambda: (b80, lambda: N.mlp_half_wide_forward(x80, *mlp_args(b80), 78)),
    "k_mlp_fwd_w (C=160)": lambda: (b160, lambda: N.mlp_half_wide_forward(x160, *mlp_args(b160), 156)),
    "k_attn_fwd<32>": lambda: (b32, lambda: N.attn_half_fused_forward(x32, *attn_args(b32, 1))[0]),
    "k_attn_fwd<64>": lambda: (b64, lambda: N.attn_half_fused_forward(x64, *attn_args(b64, 2))[0]),
}
noises = {
    "rocBLAS bf16 GEMM": lambda: torch.mm(big, big),
    "k_mlp_fwd_w (C=160)": None,          # filled below (needs its weight scope)
    "elementwise (HBM)": lambda: big.add_(0.0),
}
side = torch.cuda.Stream()
for cname, mk in cands.items():
    blk, fn = mk()
    with torch.no_grad(), N.use_weights(blk):
        ref = fn().clone()
        torch.cuda.synchronize()
        again = fn()
        torch.cuda.synchronize()
        line = f"{cname:22s} alone twice identical = {torch.equal(ref, again)}"
        for nname, noise in noises.items():
            hits, worst, nel = 0, 0.0, 0
            for _ in range(REPS):
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    if noise is None:
                        with N.use_weights(b160):
                            for _k in range(6):
                                N.mlp_half_wide_forward(xn, *mlp_args(b160), 156)
                    else:
                        for _k in range(6):
                            noise()
                y = fn()
                torch.cuda.synchronize()
                if not torch.equal(y, ref):
                    hits += 1
                    d = (y.float() - ref.float()).abs()
                    worst = max(worst, float(d.max()))
                    nel = max(nel, int((d > 0).sum()))
            line += f" | with {nname}: {hits}/{REPS} differ (max |d| {worst:.3g}, up to {nel} elements)"
        print(line)
