"""Parameter containers + forward walkers shared by the HRFormer and HRNet backbones.

The containers exist to give `state_dict()` the reference's key grammar so reference
checkpoints load unchanged; torch.nn leaf modules are used as *parameter holders only* (their own forward is never
called).  All compute goes through `nnops` on NHWC-in-memory bf16 maps.
"""
import torch
import torch.nn as nn

from .. import dispatch as nnops


def conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride, k // 2, bias=bias)


def conv_bn(cin, cout, k, stride=1):
    """ModuleList([conv, bn]) -> keys '<name>.0.weight', '<name>.1.{weight,bias,running_*}'."""
    return nn.ModuleList([conv(cin, cout, k, stride), nn.BatchNorm2d(cout)])


class Residual(nn.Module):
    """Parameters of a BasicBlock (hrnet.py:12-53) or Bottleneck (hrnet.py:56-103 == hrformer.py:296-344)."""

    def __init__(self, cin, planes, bottleneck, project=False):
        super().__init__()
        self.bottleneck = bottleneck
        if bottleneck:
            self.conv1, self.bn1 = conv(cin, planes, 1), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = conv(planes, planes, 3), nn.BatchNorm2d(planes)
            self.conv3, self.bn3 = conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
            if project:
                self.downsample = conv_bn(cin, planes * 4, 1)
        else:
            self.conv1, self.bn1 = conv(cin, planes, 3), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = conv(planes, planes, 3), nn.BatchNorm2d(planes)

    def forward(self, x):
        """Internal NHWC bf16 map in/out when called by a backbone (inside its scope); a public (B,C,H,W) tensor when called on its
        own, like the reference's BasicBlock / Bottleneck (hrnet.py:12-103)."""
        if not nnops._ACTIVE:
            with nnops.scope(self):
                return nnops.to_public(self.forward(nnops.from_public(x)))
        tr = self.training
        if not self.bottleneck:
            return nnops.residual_block(x, (self.conv1, self.bn1), [], (self.conv2, self.bn2), tr)
        if not hasattr(self, "downsample"):
            return nnops.residual_block(x, (self.conv1, self.bn1), [(self.conv2, self.bn2)], (self.conv3, self.bn3), tr)
        y = nnops.conv_bn_act(x, self.conv1, self.bn1, True, None, tr)
        y = nnops.conv_bn_act(y, self.conv2, self.bn2, True, None, tr)
        res = nnops.conv_bn_act(x, self.downsample[0], self.downsample[1], False, None, tr)
        return nnops.conv_bn_act(y, self.conv3, self.bn3, True, res, tr)


def make_fuse_layers(channels):
    """fuse_layers.{i}.{j}: [conv1x1,bn] for j>i, chain of (i-j) [conv3x3 s2, bn] for j<i, nothing for j==i."""
    n = len(channels)
    outer = nn.ModuleDict()
    for i in range(n):
        row = nn.ModuleDict()
        for j in range(n):
            if j > i:
                row[str(j)] = conv_bn(channels[j], channels[i], 1)
            elif j < i:
                row[str(j)] = nn.ModuleList([conv_bn(channels[j], channels[i] if s == i - j - 1 else channels[j], 3, 2)
                                             for s in range(i - j)])
        outer[str(i)] = row
    return outer


def make_transition(pre, cur):
    """transition{t}.{b}: [conv3x3,bn] when the channel count changes; chains of stride-2 convs for new branches."""
    tr = nn.ModuleDict()
    for i, c in enumerate(cur):
        if i < len(pre):
            if c != pre[i]:
                tr[str(i)] = conv_bn(pre[i], c, 3)
        else:
            tr[str(i)] = nn.ModuleList([conv_bn(pre[-1] if s == 0 else c, c, 3, 2) for s in range(i + 1 - len(pre))])
    return tr


def transition_branch(tr, i, t, n_prev, training):
    """Branch i of a transition applied to its source tensor t (= ys[min(i, n_prev - 1)]): identity, a 3x3 conv + BN when the channel
    count changes, or the chain of stride-2 convs that creates a new branch from the last one.  Called from inside the first
    module's branch task i of the new stage (no separate serial pass over the transition convs)."""
    key = str(i)
    if i < n_prev:
        return nnops.conv_bn_act(t, tr[key][0], tr[key][1], True, None, training) if key in tr else t
    for cv, bn in tr[key]:
        t = nnops.conv_bn_act(t, cv, bn, True, None, training)
    return t


def run_transition(tr, ys, n_cur, training):
    outs = []
    for i in range(n_cur):
        key = str(i)
        if i < len(ys):
            outs.append(nnops.conv_bn_act(ys[i], tr[key][0], tr[key][1], True, None, training) if key in tr else ys[i])
        else:
            t = ys[-1]
            for cv, bn in tr[key]:
                t = nnops.conv_bn_act(t, cv, bn, True, None, training)
            outs.append(t)
    return outs


def init_backbone_weights(module):
    """hrformer.py:709-720 / hrnet.py:386-393: conv kaiming-normal(fan_out, relu); norm 1/0; linear trunc-normal .02."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
