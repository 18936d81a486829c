"""Network op layer used by models/*: every op takes/returns PHYSICALLY-NHWC bf16 feature maps.

Feature maps travel as torch tensors with logical shape (B,C,H,W) in `channels_last` memory format, i.e. the
bytes in HBM are NHWC; `x.permute(0,2,3,1)` is the free (B,H,W,C) view the HIP kernels index.

`IMPL` records, per op, whether it runs as a hand-written HIP kernel ("hip") or is still an ATen composite
("aten": MIOpen/hipBLASLt/elementwise kernels dispatched by PyTorch-ROCm).  bench.py prints this table with
every result so a number is never mistaken for an all-HIP path.  ATen entries are stop-gaps to be replaced row
by row; none of them is a CPU fallback and none touches oracle/.
"""
import torch
import torch.nn.functional as F

IMPL = {
    "conv_bn_act": "aten",        # A5/A6/H1: 3x3 / 1x1 conv + BatchNorm (+residual) (+ReLU)
    "window_block": "aten",       # A1-A3: LN1 + window MSA + residual + LN2 + MLP + residual
    "exchange": "aten",           # A4: 1x1conv+BN+bilinear-up / strided 3x3 chains, sum, ReLU
    "head_out": "aten",           # H1/H2: final 1x1 conv with bias (+softplus) -> fp32 NCHW
}

ACT_DTYPE = torch.bfloat16
WS = 7


def to_features(x_nchw_f32):
    """Model input (B,3,H,W) fp32 NCHW -> bf16 NHWC-in-memory."""
    return x_nchw_f32.to(ACT_DTYPE).contiguous(memory_format=torch.channels_last)


def _bn(y, bn, training):
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, 0.1, 1e-5)


def conv_bn_act(x, conv, bn=None, relu=False, residual=None, training=False):
    y = F.conv2d(x, conv.weight.to(ACT_DTYPE), None, conv.stride, conv.padding)
    if bn is not None:
        y = _bn(y, bn, training)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def head_out(x, conv, softplus=False):
    """Final 1x1 conv with bias; fp32 NCHW output (these maps feed the fp32 loss / decoders)."""
    y = F.conv2d(x, conv.weight.to(ACT_DTYPE), conv.bias.to(ACT_DTYPE)).float().contiguous()
    return F.softplus(y) if softplus else y


def rel_bias(attn, heads):
    n = WS * WS
    return attn.relative_position_bias_table[attn.relative_position_index.reshape(-1)].reshape(n, n, heads).permute(2, 0, 1)


def window_block(x, blk, heads, scale1=None, scale2=None):
    """One HRFormer block on a (B,C,H,W) channels_last bf16 map. scale*: per-sample DropPath multipliers (B,) or None."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1)                                   # (B,H,W,C) view
    u = F.layer_norm(t.float(), (C,), blk.norm1.weight, blk.norm1.bias, 1e-5).to(ACT_DTYPE)
    Hp, Wp = -(-H // WS) * WS, -(-W // WS) * WS
    u = F.pad(u, (0, 0, 0, Wp - W, 0, Hp - H))                  # zero tokens AFTER LN1, no mask (reference semantics)
    nh, nw = Hp // WS, Wp // WS
    tok = u.reshape(B, nh, WS, nw, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nh * nw, WS * WS, C)
    a = blk.attn
    d = C // heads
    qkv = F.linear(tok, a.qkv.weight.to(ACT_DTYPE), a.qkv.bias.to(ACT_DTYPE)).reshape(-1, WS * WS, 3, heads, d).permute(2, 0, 3, 1, 4)
    bias = rel_bias(a, heads).to(ACT_DTYPE).unsqueeze(0)
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=bias, scale=d ** -0.5)
    o = o.transpose(1, 2).reshape(-1, WS * WS, C)
    o = F.linear(o, a.proj.weight.to(ACT_DTYPE), a.proj.bias.to(ACT_DTYPE))
    o = o.reshape(B, nh, nw, WS, WS, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)[:, :H, :W]
    if scale1 is not None:
        o = o * scale1.view(B, 1, 1, 1).to(ACT_DTYPE)
    t = t + o
    v = F.layer_norm(t.float(), (C,), blk.norm2.weight, blk.norm2.bias, 1e-5).to(ACT_DTYPE)
    m = F.linear(F.gelu(F.linear(v, blk.mlp.fc1.weight.to(ACT_DTYPE), blk.mlp.fc1.bias.to(ACT_DTYPE))),
                 blk.mlp.fc2.weight.to(ACT_DTYPE), blk.mlp.fc2.bias.to(ACT_DTYPE))
    if scale2 is not None:
        m = m * scale2.view(B, 1, 1, 1).to(ACT_DTYPE)
    return (t + m).permute(0, 3, 1, 2)                          # logical NCHW, still NHWC in memory


def exchange(xs, fuse, training, n_out=None):
    """Exchange unit: out_i = relu(sum_j route_{j->i}(x_j)), j ascending (hrformer.py:462-491 == hrnet.py:198-227)."""
    n = len(xs)
    outs = []
    for i in range(n if n_out is None else n_out):
        acc = None
        for j in range(n):
            if j == i:
                t = xs[j]
            elif j > i:
                conv, bn = fuse[str(i)][str(j)]
                t = conv_bn_act(xs[j], conv, bn, False, None, training)
                t = F.interpolate(t, size=xs[i].shape[-2:], mode="bilinear", align_corners=False)
            else:
                t = xs[j]
                chain = fuse[str(i)][str(j)]
                for s, (conv, bn) in enumerate(chain):
                    t = conv_bn_act(t, conv, bn, s != len(chain) - 1, None, training)
            acc = t if acc is None else acc + t
        outs.append(F.relu(acc))
    return outs


def drop_scales(n_draws, batch, drop_prob, device):
    """All DropPath multipliers of one step in one launch: floor(keep + U)/keep, shape (n_draws, B)."""
    keep = 1.0 - drop_prob
    return torch.floor(keep + torch.rand(n_draws, batch, device=device)) / keep
