"""Oracle (test infrastructure only): the networks as pure functions over a flat state dict.

Restates, in torch-CPU fp32 (or fp64), what the reference's modules compute:
A1-A3 models/hrformer.py:67-293 · A4 hrformer.py:420-491 == hrnet.py:157-227 · A5 hrnet.py:12-103,
hrformer.py:296-344 · A6/A7 hrformer.py:548-776, hrnet.py:243-441 · H1 fusion_head.py:195-307 ·
H2 pose_estimator.py:22-99.  Parameters are looked up by the reference's state_dict keys
 in a plain {key: tensor} dict `P`; nothing here is an nn.Module.
Pinned by tests/golden/{attn_blocks,modules,head_loss,model_level}.npz.
"""
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

WS = 7  # window size of every HRFormer stage (hrformer.py:522)


@dataclass
class Ctx:
    """Evaluation context: BN mode + running-stat capture, DropPath scales, optional rounding hook."""
    train: bool = False
    bn_updates: Dict[str, tuple] = field(default_factory=dict)
    drop_scale: Optional[Callable[[str, int], Optional[torch.Tensor]]] = None  # (block key, which) -> (B,) scale or None
    q: Callable[[torch.Tensor], torch.Tensor] = staticmethod(lambda t: t)       # storage rounding emulation (e.g. bf16)


def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _StorageRound(torch.autograd.Function):
    """bf16 storage emulation of one tensor: the value is rounded on the way forward and its gradient on the way back (the HIP path
    keeps every activation AND every activation gradient in bf16 between kernels)."""

    @staticmethod
    def forward(ctx, t):
        return bf16_round(t)

    @staticmethod
    def backward(ctx, g):
        return bf16_round(g)


def bf16_storage(t):
    return _StorageRound.apply(t) if t.requires_grad else bf16_round(t)


def bf16_weights(P):
    """The compute copies of the HIP path: conv / linear weights rounded to bf16 (with a straight-through gradient so the result can be
    differentiated w.r.t. the fp32 masters); biases, norm affines, rel-pos tables and buffers stay fp32."""
    out = {}
    for k, v in P.items():
        if v.is_floating_point() and v.dim() in (2, 4) and k.endswith(".weight"):
            out[k] = v + (bf16_round(v.detach()) - v.detach())
        else:
            out[k] = v
    return out


# ------------------------------------------------------------------------------ conv / BN pieces
def conv(x, P, key, stride=1, pad=None):
    w = P[key + ".weight"]
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, P.get(key + ".bias"), stride=stride, padding=pad)


def batchnorm(x, P, key, ctx):
    """Train: biased batch variance normalises, running_var gets the unbiased one, momentum 0.1, eps 1e-5.
    With a storage-rounding hook (ctx.q) the statistics are those of the fp32 conv output while the normalised tensor is the
    rounded one -- the HIP path takes the sums from the GEMM's fp32 accumulators and stores the raw output in bf16."""
    g, b = P[key + ".weight"], P[key + ".bias"]
    xs = ctx.q(x)
    if not ctx.train:
        mean, var = P[key + ".running_mean"].detach(), P[key + ".running_var"].detach()
        return (xs - mean[None, :, None, None]) * torch.rsqrt(var + 1e-5)[None, :, None, None] * g[None, :, None, None] + b[None, :, None, None]
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    n = x.numel() // x.shape[1]
    with torch.no_grad():
        ctx.bn_updates[key] = (0.9 * P[key + ".running_mean"] + 0.1 * mean,
                               0.9 * P[key + ".running_var"] + 0.1 * var * (n / max(n - 1, 1)))
    return (xs - mean[None, :, None, None]) * torch.rsqrt(var + 1e-5)[None, :, None, None] * g[None, :, None, None] + b[None, :, None, None]


def conv_bn(x, P, ckey, bkey, ctx, stride=1, relu=False):
    y = batchnorm(conv(x, P, ckey, stride), P, bkey, ctx)
    return ctx.q(torch.relu(y) if relu else y)


def basic_block(x, P, pre, ctx):
    y = conv_bn(x, P, pre + ".conv1", pre + ".bn1", ctx, relu=True)
    y = batchnorm(conv(y, P, pre + ".conv2"), P, pre + ".bn2", ctx)
    return ctx.q(torch.relu(y + x))


def bottleneck(x, P, pre, ctx):
    y = conv_bn(x, P, pre + ".conv1", pre + ".bn1", ctx, relu=True)
    y = conv_bn(y, P, pre + ".conv2", pre + ".bn2", ctx, relu=True)
    y = batchnorm(conv(y, P, pre + ".conv3"), P, pre + ".bn3", ctx)
    res = x
    if pre + ".downsample.0.weight" in P:
        res = ctx.q(batchnorm(conv(x, P, pre + ".downsample.0"), P, pre + ".downsample.1", ctx))
    return ctx.q(torch.relu(y + res))


def upsample_bilinear(x, size):
    """F.interpolate(..., mode='bilinear', align_corners=False) spelled out: src=(dst+.5)*in/out-.5, clamped at 0."""
    B, C, H, W = x.shape

    def taps(n_in, n_out):
        s = (torch.arange(n_out, dtype=x.dtype) + 0.5) * (n_in / n_out) - 0.5
        s = s.clamp(min=0)
        i0 = s.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, s - i0

    y0, y1, fy = taps(H, size[0])
    x0, x1, fx = taps(W, size[1])
    rows = x[:, :, y0] * (1 - fy)[None, None, :, None] + x[:, :, y1] * fy[None, None, :, None]
    return rows[..., x0] * (1 - fx) + rows[..., x1] * fx


def exchange(xs, P, pre, ctx, n_out=None):
    """Multi-resolution fusion (exchange unit): out_i = relu( Σ_j route_{j->i}(x_j) ), j ascending."""
    n = len(xs)
    outs = []
    for i in range(n if n_out is None else n_out):
        acc = None
        for j in range(n):
            if j == i:
                t = xs[j]
            elif j > i:
                t = ctx.q(batchnorm(conv(xs[j], P, f"{pre}.{i}.{j}.0"), P, f"{pre}.{i}.{j}.1", ctx))     # stored, then up-sampled inside the sum
                t = upsample_bilinear(t, xs[i].shape[-2:])
            else:
                t = xs[j]
                for s in range(i - j):
                    t = conv_bn(t, P, f"{pre}.{i}.{j}.{s}.0", f"{pre}.{i}.{j}.{s}.1", ctx, stride=2, relu=(s != i - j - 1))
            acc = t if acc is None else acc + t
        outs.append(ctx.q(torch.relu(acc)))
    return outs


# ------------------------------------------------------------------------------ transformer pieces
def to_windows(x, ws=WS):
    """(B,H,W,C) -> (B*nH*nW, ws*ws, C), zero-padding bottom/right up to a multiple of ws."""
    B, H, W, C = x.shape
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    canvas = x.new_zeros(B, Hp, Wp, C)
    canvas[:, :H, :W] = x
    t = canvas.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return t.reshape(-1, ws * ws, C), (Hp, Wp)


def from_windows(t, B, H, W, Hp, Wp, ws=WS):
    C = t.shape[-1]
    canvas = t.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    return canvas[:, :H, :W]


def rel_bias(P, pre, heads, ws=WS):
    """(heads, N, N): table[(dy+ws-1)*(2ws-1) + (dx+ws-1), h]."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    idx = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
    table = P.get(pre + ".relative_position_bias_table")
    if table is None:                    # with_rpe=False (hrformer.py:186-191): no bias term
        return torch.zeros(heads, ws * ws, ws * ws)
    return table[idx.reshape(-1)].reshape(ws * ws, ws * ws, heads).permute(2, 0, 1)


def window_attention(tok, P, pre, heads, ctx=None):
    """(Bw,N,C) -> (Bw,N,C): qkv, q*d^-.5, QK^T + bias, softmax, AV, proj. No mask: pad tokens attend."""
    q_ = ctx.q if ctx is not None else (lambda t: t)
    Bw, N, C = tok.shape
    d = C // heads
    qkv = q_(tok @ P[pre + ".qkv.weight"].T + P[pre + ".qkv.bias"]).reshape(Bw, N, 3, heads, d)
    q, k, v = qkv[:, :, 0] * d ** -0.5, qkv[:, :, 1], qkv[:, :, 2]
    logits = torch.einsum("bnhd,bmhd->bhnm", q, k) + rel_bias(P, pre, heads)[None]
    p = q_(torch.softmax(logits, -1))
    o = q_(torch.einsum("bhnm,bmhd->bnhd", p, v).reshape(Bw, N, C))
    return o @ P[pre + ".proj.weight"].T + P[pre + ".proj.bias"]


def hrformer_block(x, P, pre, heads, ctx):
    """NHWC in/out. x + dp*attn(LN1(x)); then + dp*fc2(gelu(fc1(LN2(.))))."""
    B, H, W, C = x.shape
    u = ctx.q(F.layer_norm(x, (C,), P[pre + ".norm1.weight"], P[pre + ".norm1.bias"], 1e-5))
    tok, (Hp, Wp) = to_windows(u)
    a = from_windows(window_attention(tok, P, pre + ".attn", heads, ctx), B, H, W, Hp, Wp)
    s1 = ctx.drop_scale(pre, 0) if ctx.drop_scale else None
    x = ctx.q(x + (a if s1 is None else a * s1.view(B, 1, 1, 1)))
    v = ctx.q(F.layer_norm(x, (C,), P[pre + ".norm2.weight"], P[pre + ".norm2.bias"], 1e-5))
    hdn = ctx.q(F.gelu(v @ P[pre + ".mlp.fc1.weight"].T + P[pre + ".mlp.fc1.bias"]))
    m = hdn @ P[pre + ".mlp.fc2.weight"].T + P[pre + ".mlp.fc2.bias"]
    s2 = ctx.drop_scale(pre, 1) if ctx.drop_scale else None
    return ctx.q(x + (m if s2 is None else m * s2.view(B, 1, 1, 1)))


def _count(P, pre):
    n = 0
    while any(k.startswith(f"{pre}.{n}.") for k in P):
        n += 1
    return n


def hrformer_module(xs, P, pre, heads, ctx):
    nb = len(xs)
    ys = []
    for b in range(nb):
        t = xs[b].permute(0, 2, 3, 1)
        for blk in range(_count(P, f"{pre}.branches.{b}")):
            t = hrformer_block(t, P, f"{pre}.branches.{b}.{blk}", heads[b], ctx)
        ys.append(t.permute(0, 3, 1, 2))
    return ys if nb == 1 else exchange(ys, P, pre + ".fuse_layers", ctx)


def hrnet_module(xs, P, pre, ctx):
    nb = len(xs)
    ys = []
    for b in range(nb):
        t = xs[b]
        for blk in range(_count(P, f"{pre}.branches.{b}")):
            t = basic_block(t, P, f"{pre}.branches.{b}.{blk}", ctx)
        ys.append(t)
    return ys if nb == 1 else exchange(ys, P, pre + ".fuse_layers", ctx)


def transition(ys, P, pre, n_cur, ctx):
    outs = []
    for i in range(n_cur):
        if i < len(ys):
            outs.append(conv_bn(ys[i], P, f"{pre}.{i}.0", f"{pre}.{i}.1", ctx, relu=True) if f"{pre}.{i}.0.weight" in P else ys[i])
        else:
            t = ys[-1]
            for s in range(i + 1 - len(ys)):
                t = conv_bn(t, P, f"{pre}.{i}.{s}.0", f"{pre}.{i}.{s}.1", ctx, stride=2, relu=True)
            outs.append(t)
    return outs


def backbone(x, P, ctx, pre="backbone"):
    """HRFormer or HRNet, decided by the keys present. Returns branch-0 feature (B,C0,H/4,W/4)."""
    is_former = any(".attn.qkv.weight" in k for k in P if k.startswith(pre))
    x = ctx.q(x)                        # the HIP path converts the image to bf16 NHWC once
    x = conv_bn(x, P, pre + ".conv1", pre + ".bn1", ctx, stride=2, relu=True)
    x = conv_bn(x, P, pre + ".conv2", pre + ".bn2", ctx, stride=2, relu=True)
    for i in range(_count(P, pre + ".layer1")):
        x = bottleneck(x, P, f"{pre}.layer1.{i}", ctx)
    ys = [x]
    for stage in (2, 3, 4):
        ys = transition(ys, P, f"{pre}.transition{stage - 1}", stage, ctx)
        for m in range(_count(P, f"{pre}.stage{stage}")):
            mp = f"{pre}.stage{stage}.{m}"
            if is_former:
                heads = [P[f"{mp}.branches.{b}.0.attn.relative_position_bias_table"].shape[1] for b in range(stage)]
                ys = hrformer_module(ys, P, mp, heads, ctx)
            else:
                ys = hrnet_module(ys, P, mp, ctx)
    return ys[0]


# ------------------------------------------------------------------------------ heads
def fusion_head(x, P, ctx, pre="head"):
    f = conv_bn(x, P, pre + ".shared_layers.0", pre + ".shared_layers.1", ctx, relu=True)
    f = conv_bn(f, P, pre + ".shared_layers.3", pre + ".shared_layers.4", ctx, relu=True)

    def branch(name):
        t = conv_bn(f, P, f"{pre}.{name}.0", f"{pre}.{name}.1", ctx, relu=True)
        return conv(t, P, f"{pre}.{name}.3", pad=0)

    hm = branch("heatmap_branch")
    off = branch("offset_branch")
    var = F.softplus(branch("variance_branch"))
    B, _, H, W = off.shape
    return {"heatmaps": hm, "offsets": off.reshape(B, -1, 2, H, W), "variances": var,
            "fusion_weight": torch.sigmoid(P[pre + ".fusion_weight"])}


def heatmap_head(x, P, pre="head"):
    return conv(x, P, pre + ".final_layer", pad=0)


def pose_forward(x, P, ctx):
    feat = backbone(x, P, ctx)
    if "head.fusion_weight" in P:
        return fusion_head(feat, P, ctx)
    return {"heatmaps": heatmap_head(feat, P)}


def apply_bn_updates(P, ctx):
    for key, (m, v) in ctx.bn_updates.items():
        P[key + ".running_mean"], P[key + ".running_var"] = m, v
        P[key + ".num_batches_tracked"] = P[key + ".num_batches_tracked"] + 1
