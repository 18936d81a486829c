#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the shapes of HRFormer-small, B=64, 256x192 (run on the GPU box).

    python scripts/bench_kernels.py [filter]

Prints one line per case: average device time (HIP events on the launch stream), achieved TFLOP/s or GB/s, and the
fraction of the MI355X peak (2.5 PFLOP/s dense bf16, 8 TB/s HBM).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infantposeestimation_gaussianbias_amd import nnops  # noqa: E402
from infantposeestimation_gaussianbias_amd._lib import call, lib, stream_ptr  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, sec, flops=None, bytes_=None):
    s = f"{name:58s} {sec * 1e6:9.1f} us"
    if flops:
        s += f"  {flops / sec / 1e12:8.1f} TFLOP/s ({flops / sec / 2.5e15 * 100:5.1f}% mfma)"
    if bytes_:
        s += f"  {bytes_ / sec / 1e9:8.1f} GB/s ({bytes_ / sec / 8e12 * 100:5.1f}% hbm)"
    print(s, flush=True)


class Holder(torch.nn.Module):
    def __init__(self, **m):
        super().__init__()
        for k, v in m.items():
            setattr(self, k, v)


def conv_cases(flt):
    B = 64
    for (H, W, Cin, Cout, k, s) in [(64, 48, 256, 256, 3, 1), (64, 48, 256, 128, 3, 1), (64, 48, 256, 32, 3, 1), (64, 48, 32, 256, 3, 1), (64, 48, 64, 64, 3, 1),
                                    (64, 48, 64, 256, 1, 1), (64, 48, 256, 64, 1, 1), (128, 96, 64, 64, 3, 2), (256, 192, 8, 64, 3, 2),
                                    (32, 24, 64, 64, 3, 1), (64, 48, 32, 64, 3, 2), (64, 48, 32, 32, 3, 1), (96, 72, 32, 32, 3, 1), (48, 36, 64, 64, 3, 1),
                                    (16, 12, 64, 64, 3, 1), (32, 24, 32, 64, 3, 1), (32, 24, 64, 32, 3, 1)]:
        name = f"conv {Cin}->{Cout} k{k} s{s} @{H}x{W}"
        if flt and flt not in name:
            continue
        conv = torch.nn.Conv2d(Cin if Cin != 8 else 3, Cout, k, s, k // 2, bias=False)
        m = Holder(c=conv).to(DEV)
        x = torch.randn(B, H, W, Cin, device=DEV).to(BF)
        with nnops.use_weights(m) as wc:
            wf, wd = wc.fwd[id(m.c.weight)], wc.dgrad[id(m.c.weight)]
            Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            M = B * Ho * Wo
            flops = 2.0 * M * Cout * Cin * k * k
            sec = timeit(lambda: nnops._conv_raw(x, wf, Cout, k, s, True))
            report(name + " fwd+stats", sec, flops, (x.numel() + M * Cout) * 2)
            g = torch.randn(B, Ho, Wo, Cout, device=DEV).to(BF)
            if Cin != 8:
                sec = timeit(lambda: nnops._conv_dgrad(g, wd, Cin, k, s, (H, W)))
                report(name + " dgrad", sec, flops)
            sec = timeit(lambda: nnops._wgrad(x, g, Cout, Cin, k, s, (B, H, W, Ho, Wo)))
            report(name + " wgrad", sec, flops)


def linear_cases(flt):
    for (M, K, N, act, tag) in [(219520, 32, 96, 0, "qkv b0"), (196608, 32, 128, 1, "fc1 b0"), (196608, 128, 32, 0, "fc2 b0"),
                                (49152, 64, 256, 1, "fc1 b1"), (49152, 256, 64, 0, "fc2 b1"), (12288, 128, 512, 1, "fc1 b2"),
                                (3072, 256, 1024, 1, "fc1 b3"), (3072, 1024, 256, 0, "fc2 b3")]:
        name = f"linear {tag} M={M} K={K} N={N}"
        if flt and flt not in name:
            continue
        x = torch.randn(M, K, device=DEV).to(BF)
        w = torch.randn(N, K, device=DEV).to(BF)
        bias = torch.randn(N, device=DEV)
        z = torch.empty(M, N, device=DEV, dtype=BF) if act else None
        sec = timeit(lambda: nnops._linear(x, w, M, N, K, bias=bias, preact=z, act=act))
        report(name + (" +gelu" if act else ""), sec, 2.0 * M * N * K, (M * K + M * N * (2 if act else 1)) * 2)
        g = torch.randn(M, N, device=DEV).to(BF)
        sec = timeit(lambda: nnops._wgrad(x, g, N, K, 1, 1, None, M=M))
        report(name + " wgrad", sec, 2.0 * M * N * K, (M * K + M * N) * 2)
        sec = timeit(lambda: nnops._colsum(g, M, N))
        report(name + " colsum", sec, None, M * N * 2)


def attn_cases(flt):
    for (nw, heads, C) in [(4480, 1, 32), (1280, 2, 64), (384, 4, 128), (128, 8, 256)]:
        name = f"attn nw={nw} h={heads} C={C}"
        if flt and flt not in name:
            continue
        qkv = torch.randn(nw * 49, 3 * C, device=DEV).to(BF)
        table = torch.randn(169, heads, device=DEV)
        o = torch.empty(nw * 49, C, device=DEV, dtype=BF)
        lse = torch.empty(nw * heads * 49, device=DEV)
        flops = 4.0 * nw * heads * 49 * 49 * (C // heads)
        sec = timeit(lambda: call("pk_window_attn_fwd", qkv, table, o, lse, nw, heads, C, 0.0, stream_ptr()))
        report(name + " fwd", sec, flops, (qkv.numel() + o.numel()) * 2)
        go = torch.randn(nw * 49, C, device=DEV).to(BF)
        dqkv = torch.empty_like(qkv)
        part = torch.empty(lib.pk_window_attn_bwd_ws_floats(nw, heads), device=DEV)
        dt = torch.empty(169, heads, device=DEV)
        sec = timeit(lambda: call("pk_window_attn_bwd", qkv, table, o, go, lse, dqkv, part, dt, nw, heads, C, 0.0, stream_ptr()))
        report(name + " bwd", sec, 2.5 * flops, (2 * qkv.numel() + go.numel()) * 2)


def elementwise_cases(flt):
    M, C = 196608, 256
    name = "bn_act 196608x256"
    if not flt or flt in name:
        x = torch.randn(M, C, device=DEV).to(BF)
        y = torch.empty_like(x)
        sc, sh = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        sec = timeit(lambda: call("pk_bn_act", x, sc, sh, None, y, M, C, 1, None, stream_ptr()))
        report(name, sec, None, 2 * M * C * 2)
    for Cc in (32, 256):
        name = f"layernorm fwd 196608x{Cc}"
        if flt and flt not in name:
            continue
        x = torch.randn(M, Cc, device=DEV).to(BF)
        g, b = torch.randn(Cc, device=DEV), torch.randn(Cc, device=DEV)
        sec = timeit(lambda: nnops._layernorm(x, g, b))
        report(name, sec, None, 2 * M * Cc * 2)


def block_cases(flt):
    """HRFormer block halves at the four branch shapes of HRFormer-small (B = 64): fused kernels vs the unfused sequences.
    Algorithmic bytes: forward reads x and writes y (4C B/token); backward reads x, dy and writes dx (6C B/token)."""
    from infantposeestimation_gaussianbias_amd.models.hrformer import HRFormerBlock
    for (H, W, C, heads) in [(64, 48, 32, 1), (32, 24, 64, 2), (16, 12, 128, 4), (8, 6, 256, 8)]:
        name = f"block C={C} @{H}x{W}"
        if flt and flt not in name:
            continue
        B = 64
        blk = HRFormerBlock(C, heads).to(DEV)
        x = torch.randn(B, H, W, C, device=DEV).to(BF).requires_grad_(True)
        gy = torch.randn(B, H, W, C, device=DEV).to(BF)
        s2 = torch.ones(B, device=DEV)
        M = B * H * W
        m = blk.mlp
        margs = (blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, s2)
        with nnops.use_weights(blk) as wc:
            variants = [("mlp unfused", lambda: nnops._MlpHalf.apply(x, *margs, 0))]
            if nnops.fused_mlp_enabled(C):
                variants.append(("mlp FUSED", lambda: nnops._MlpHalfFused.apply(x, *margs)))
            for tag, fn in variants:
                with torch.no_grad():
                    sec = timeit(fn)
                report(f"{name} {tag} fwd", sec, 16.0 * M * C * C, 4 * M * C)
                y = fn()
                sec = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
                report(f"{name} {tag} bwd (all launches)", sec, 32.0 * M * C * C, 6 * M * C)
            if nnops.fused_mlp_enabled(C):
                w1f, w1t, w2t = wc.fwd[id(m.fc1.weight)], wc.dgrad[id(m.fc1.weight)], wc.dgrad[id(m.fc2.weight)]
                dx = torch.empty_like(x)
                lnp = torch.empty(lib.pk_ln_mlp_dx_blocks(M, C) * 2 * C, device=DEV)
                slabs = torch.empty((4 * C // lib.pk_ln_mlp_hidden_slice(C)) * lib.pk_ln_mlp_dw_blocks(M, C) * lib.pk_ln_mlp_slab_floats(C), device=DEV)
                xd = x.detach()
                sec = timeit(lambda: call("pk_ln_mlp_bwd_dx", gy, xd, blk.norm2.weight, blk.norm2.bias, w1f, m.fc1.bias, w1t, w2t, s2, dx, lnp,
                                          M, C, H * W, 1e-5, stream_ptr()))
                report(f"{name} mlp FUSED bwd_dx kernel", sec, 24.0 * M * C * C, 6 * M * C)
                sec = timeit(lambda: call("pk_ln_mlp_bwd_dw", gy, xd, blk.norm2.weight, blk.norm2.bias, w1f, m.fc1.bias, w2t, s2, slabs,
                                          M, C, H * W, 1e-5, stream_ptr()))
                report(f"{name} mlp FUSED bwd_dw kernel", sec, 32.0 * M * C * C, 4 * M * C + slabs.numel() * 4)
            a = blk.attn
            aargs = (blk.norm1.weight, blk.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                     s2, heads)
            variants = [("attn unfused", lambda: nnops._AttnHalf.apply(x, *aargs))]
            if nnops.fused_attn_enabled(C, heads):
                variants.append(("attn FUSED", lambda: nnops._AttnHalfFused.apply(x, *aargs)))
            for tag, fn in variants:
                with torch.no_grad():
                    sec = timeit(fn if "unfused" in tag else (lambda: nnops.attn_half_fused_forward(x, *aargs)))
                report(f"{name} {tag} fwd", sec, None, 4 * M * C)
                y = fn()
                sec = timeit(lambda: torch.autograd.grad(y, x, gy, retain_graph=True))
                report(f"{name} {tag} bwd (all launches)", sec, None, 6 * M * C)
            if nnops.fused_attn_enabled(C, heads):
                amap, nwin = nnops.window_rowmap(B, H, W, x.device)
                nw = B * nwin
                y, o, lse, _ = nnops.attn_half_fused_forward(x.detach(), *aargs, save=True)
                nb = lib.pk_attn_block_blocks(nw)
                lnp, rpb = torch.empty(nb * 2 * C, device=DEV), torch.empty(nb * 4 * heads * 169, device=DEV)
                dx, dqkv, u_w = torch.empty_like(x), torch.empty(nw * 49, 3 * C, device=DEV, dtype=BF), torch.empty(nw * 49, C, device=DEV, dtype=BF)
                sec = timeit(lambda: call("pk_attn_block_bwd", gy, x.detach(), amap, blk.norm1.weight, blk.norm1.bias, a.relative_position_bias_table,
                                          wc.fwd[id(a.qkv.weight)], a.qkv.bias, wc.dgrad[id(a.qkv.weight)], wc.dgrad[id(a.proj.weight)], s2, o, lse,
                                          dx, dqkv, u_w, lnp, rpb, nw, nwin, heads, C, 1e-5, stream_ptr()))
                report(f"{name} attn FUSED bwd kernel", sec, None, 6 * M * C)


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else None
    print(torch.cuda.get_device_name(0))
    conv_cases(flt)
    linear_cases(flt)
    attn_cases(flt)
    elementwise_cases(flt)
    block_cases(flt)
