"""HRNet backbone (drop-in for the reference's models/hrnet.py: `HRNet(base_channels)`, `hrnet_w32/w48`).

Stem -> 4 bottlenecks -> stage2 (1 module) -> stage3 (4 modules) -> stage4 (3 modules), 4 BasicBlocks per branch per
module, exchange unit after each module (hrnet.py:243-302); returns branch 0 only (hrnet.py:441).
"""
import os

import torch
import torch.nn as nn

from .. import dispatch as nnops
from ._blocks import Residual, conv, init_backbone_weights, make_fuse_layers, make_transition, run_transition, transition_branch


class HighResolutionModule(nn.Module):
    def __init__(self, channels, blocks_per_branch):
        super().__init__()
        self.branches = nn.ModuleList(nn.ModuleList(Residual(c, c, False) for _ in range(nb))
                                      for c, nb in zip(channels, blocks_per_branch))
        if len(channels) > 1:
            self.fuse_layers = make_fuse_layers(channels)

    def forward(self, xs, pre=None, defer=False, trans=None, first_only=False):
        """`pre` / `defer` / `trans`: chained modules, see HRFormerModule.forward (the previous module's exchange output i, or branch i
        of the stage's transition, is computed inside this module's branch task i)."""
        n_prev = len(xs)

        def make(b, blocks):
            def run(ins):
                t = nnops.exchange_output(b, ins, pre, self.training) if pre is not None else ins[0]
                if trans is not None:
                    t = transition_branch(trans, b, t, n_prev, self.training)
                for blk in blocks:
                    t = blk(t)
                return t
            return run

        ys = nnops.parallel([make(b, blocks) for b, blocks in enumerate(self.branches)],
                            [(list(xs) if pre is not None else [xs[min(b, n_prev - 1)]]) for b in range(len(self.branches))])
        return ys if (len(ys) == 1 or defer) else nnops.exchange(ys, self.fuse_layers, self.training, first_only=first_only)


_CHAIN = os.environ.get("POSE_CHAIN_MODULES", "1") != "0"


class HRNet(nn.Module):
    def __init__(self, in_channels: int = 3, base_channels: int = 32, _stage_channels=None):
        """`_stage_channels` (4 ints) overrides base_channels * 2^i: used for 8-aligned padded twins (models/padded.py)."""
        super().__init__()
        c = self.base_channels = base_channels
        self.in_channels = in_channels
        self.stage_channels = list(_stage_channels) if _stage_channels is not None else [c * (2 ** i) for i in range(4)]
        self.conv1, self.bn1 = conv(in_channels, 64, 3, 2), nn.BatchNorm2d(64)
        self.conv2, self.bn2 = conv(64, 64, 3, 2), nn.BatchNorm2d(64)
        self.layer1 = nn.ModuleList(Residual(64 if i == 0 else 256, 64, True, project=(i == 0)) for i in range(4))
        pre = [256]
        for s, nm in ((2, 1), (3, 4), (4, 3)):
            ch = self.stage_channels[:s]
            setattr(self, f"transition{s - 1}", make_transition(pre, ch))
            setattr(self, f"stage{s}", nn.ModuleList(HighResolutionModule(ch, [4] * s) for _ in range(nm)))
            pre = ch
        self.out_channels = self.stage_channels[0]
        init_backbone_weights(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tw = nnops.padded_twin(self)
        if tw is not None:                       # C % 8 != 0 (HRNet-W18): run the 8-aligned twin, hand back the real channels
            return tw.run(x)[:, :self.out_channels]
        with nnops.scope(self):
            return nnops.to_public(self._forward(x))

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        tr = self.training
        x = nnops.to_features(x)
        x = nnops.conv_bn_act(x, self.conv1, self.bn1, True, None, tr)
        x = nnops.conv_bn_act(x, self.conv2, self.bn2, True, None, tr)
        for blk in self.layer1:
            x = blk(x)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(self, f"transition{s - 1}") if _CHAIN else None
            if not _CHAIN:
                ys = run_transition(getattr(self, f"transition{s - 1}"), ys, s, tr)
            mods_s, pre = list(getattr(self, f"stage{s}")), None
            for k, m in enumerate(mods_s):
                chain = _CHAIN and k + 1 < len(mods_s)
                ys = m(ys, pre=pre, defer=chain, trans=trans if k == 0 else None, first_only=(s == 4 and k + 1 == len(mods_s)))
                pre = m.fuse_layers if chain else None
            # N > 1: once backward has passed this boundary the later stages' gradients are exchanged while the earlier stages still
            # run backward (only output 0 of the last stage is consumed, hrformer.py:776 / hrnet.py:441)
            nnops.backward_milestone(ys if s < 4 else ys[:1])
        return ys[0]


def BasicBlock(inplanes: int, planes: int, stride: int = 1, downsample=None) -> Residual:
    """hrnet.py:12-53 (stride / downsample are never used with BasicBlock by the reference's builders)."""
    if stride != 1 or downsample is not None:
        raise ValueError("BasicBlock: stride 1 without downsample (the only form the reference builds)")
    return Residual(inplanes, planes, False)


def Bottleneck(inplanes: int, planes: int, stride: int = 1, downsample=None) -> Residual:
    """hrnet.py:56-103: expansion 4; `downsample` (any module) selects the 1x1 conv + BN projection of the identity path."""
    if stride != 1:
        raise ValueError("Bottleneck: stride 1 (the only form the reference builds)")
    return Residual(inplanes, planes, True, project=downsample is not None)


def hrnet_w32(pretrained: bool = False) -> HRNet:
    return HRNet(base_channels=32)


def hrnet_w48(pretrained: bool = False) -> HRNet:
    return HRNet(base_channels=48)


def hrnet_w18(pretrained: bool = False) -> HRNet:
    """BASELINE config 1 (`HRNet(base_channels=18)`); the reference has no named constructor for it."""
    return HRNet(base_channels=18)
