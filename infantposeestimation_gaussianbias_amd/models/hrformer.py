"""HRFormer backbone (drop-in for the reference's models/hrformer.py: same constructors, same state_dict keys).

Input (B,3,H,W) fp32 NCHW; output the branch-0 feature map (B,C0,H/4,W/4) as a channels_last bf16 tensor
(NHWC bytes).  Stage structure follows hrformer.py:505-616: stem -> 2 bottlenecks -> stage2 (1 module, 2 branches)
-> stage3 (4 modules, 3 branches) -> stage4 (2 modules, 4 branches), 2 window-attention blocks per branch,
exchange unit after every module; only output 0 of the last module is returned (hrformer.py:776).
"""
from typing import Sequence

import os

import torch
import torch.nn as nn

from .. import dispatch as nnops
from ._blocks import Residual, conv, init_backbone_weights, make_fuse_layers, make_transition, run_transition, transition_branch


def drop_path(x: torch.Tensor, drop_prob: float = 0., training: bool = False) -> torch.Tensor:
    """Stochastic depth per sample (hrformer.py:15-24): (x / keep) * floor(keep + U[0,1)) with one draw per sample; identity when
    drop_prob is 0 or not training.  The draw comes from torch's device generator, the scaling is pk_drop_path_f32."""
    if drop_prob == 0. or not training:
        return x
    from .. import hipops
    keep = 1.0 - drop_prob
    mask = torch.floor(keep + torch.rand(x.shape[0], dtype=torch.float32, device=x.device))
    return hipops.drop_path(x.float().contiguous(), mask, keep).to(x.dtype)


class DropPath(nn.Module):
    """hrformer.py:27-35.  (Inside HRFormerBlock the two draws of a block are applied in the GEMM epilogues, not through this module.)"""

    def __init__(self, drop_prob: float = 0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return drop_path(x, self.drop_prob, self.training)


def window_partition(x: torch.Tensor, window_size: int):
    """(B,H,W,C) -> ((B*nW, ws, ws, C) windows, (Hp, Wp)); zero tokens appended at the bottom / right (hrformer.py:67-91).  Inside the
    network the partition is a row map consumed by the GEMMs; this stand-alone form moves the rows with pk_rows_by_map."""
    if window_size != 7:
        raise ValueError("window size must be 7")
    from .. import hipops, nnops as _nn
    B, H, W, C = x.shape
    rowmap, nwin = _nn.window_rowmap(B, H, W, x.device)
    wins = hipops.rows_by_map(x.reshape(B * H * W, C), rowmap, rowmap.numel(), False)
    return wins.view(B * nwin, 7, 7, C), (-(-H // 7) * 7, -(-W // 7) * 7)


def window_reverse(windows: torch.Tensor, window_size: int, H: int, W: int, Hp: int, Wp: int) -> torch.Tensor:
    """(B*nW, ws, ws, C) -> (B,H,W,C), the padded rows / columns cropped (hrformer.py:94-114)."""
    if window_size != 7:
        raise ValueError("window size must be 7")
    from .. import hipops, nnops as _nn
    B = int(windows.shape[0] / (Hp * Wp / 49))
    C = windows.shape[-1]
    rowmap, _ = _nn.window_rowmap(B, H, W, windows.device)
    return hipops.rows_by_map(windows.reshape(-1, C), rowmap, B * H * W, True).view(B, H, W, C)


class WindowAttentionParams(nn.Module):
    """qkv/proj + relative-position table (169, heads) + the (49,49) index buffer (hrformer.py:147-170)."""

    def __init__(self, dim, heads, ws=7, with_rpe=True):
        super().__init__()
        self.with_rpe = with_rpe
        if with_rpe:                           # hrformer.py:148-170: without it the module holds neither the table nor the index buffer
            self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
            nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
            ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
            ys, xs = ys.reshape(-1), xs.reshape(-1)
            self.register_buffer("relative_position_index",
                                 (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.num_heads = heads

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B_*, 49, C) window tokens -> same shape (hrformer.py:174-200): qkv, scaled QK^T + relative position bias, softmax, AV, proj."""
        B_, N, C = x.shape
        if N != 49:
            raise ValueError("WindowAttention: windows of 7x7 = 49 tokens")
        with nnops.scope(self):
            return nnops.window_attention_tokens(x.reshape(B_ * N, C).to(torch.bfloat16), self, self.num_heads).view(B_, N, C)


class MlpParams(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(..., C) -> (..., C): fc2(GELU_erf(fc1(x))) (hrformer.py:38-64; dropout p = 0)."""
        with nnops.scope(self):
            return nnops.mlp_rows(x.reshape(-1, x.shape[-1]).to(torch.bfloat16), self).view(*x.shape[:-1], self.fc2.weight.shape[0])


class HRFormerBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0, drop_path=0.0, with_rpe=True):
        super().__init__()
        self.dim, self.heads, self.drop_prob = dim, heads, float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttentionParams(dim, heads, with_rpe=with_rpe)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MlpParams(dim, int(dim * mlp_ratio))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) -> (B,C,H,W) (hrformer.py:262-293): x + DropPath(attn(LN1(x))), then + DropPath(mlp(LN2(.))); DropPath draws two
        per-sample masks in training mode (hrformer.py:15-35)."""
        with nnops.scope(self):
            t = nnops.from_public(x)
            s1 = s2 = None
            if self.training and self.drop_prob > 0:
                sc = nnops.drop_scales(2, t.shape[0], self.drop_prob, t.device)
                s1, s2 = sc[0], sc[1]
            return nnops.to_public(nnops.window_block(t, self, self.heads, s1, s2))


_CHAIN = os.environ.get("POSE_CHAIN_MODULES", "1") != "0"


class HRFormerModule(nn.Module):
    def __init__(self, channels, heads, blocks_per_branch, mlp_ratios, drop_path, with_rpe=True):
        super().__init__()
        self.branches = nn.ModuleList(
            nn.ModuleList(HRFormerBlock(c, h, r, drop_path, with_rpe) for _ in range(nb))
            for c, h, nb, r in zip(channels, heads, blocks_per_branch, mlp_ratios))
        if len(channels) > 1:
            self.fuse_layers = make_fuse_layers(channels)

    def n_draws(self):
        return 2 * sum(len(b) for b in self.branches)

    def forward(self, xs, scales=None, pre=None, defer=False, trans=None, first_only=False):
        """scales: (n_draws, B) DropPath multipliers for this module (two per block, branch-major), or None.

        Chaining (same stage, same branch count): with `defer=True` the module returns its branch outputs WITHOUT the exchange unit;
        the next module gets them as `xs` together with `pre` = this module's fuse layers and computes exchange output i INSIDE its
        branch task i.  One fork/join per module instead of two, and branch i (the long high-resolution one in particular) starts
        as soon as its own input is summed instead of after the slowest exchange output (output 3: six conv + BN layers)."""
        n_prev = len(xs)

        def make(b, blocks, d0):
            def run(ins):
                t, d = (nnops.exchange_output(b, ins[:n_prev], pre, self.training) if pre is not None else ins[0]), d0
                if trans is not None:           # first module of a stage: the transition of branch b runs inside its task
                    t = transition_branch(trans, b, t, n_prev, self.training)
                k = n_prev if pre is not None else 1
                sc = ins[k] if len(ins) > k else None
                for blk in blocks:
                    s1, s2 = (sc[d], sc[d + 1]) if sc is not None else (None, None)
                    d += 2
                    t = nnops.window_block(t, blk, blk.heads, s1, s2)
                return t
            return run

        fns, d = [], 0
        for b, blocks in enumerate(self.branches):
            fns.append(make(b, blocks, d))
            d += 2 * len(blocks)
        extra = [scales] if scales is not None else []
        ys = nnops.parallel(fns, [(list(xs) if pre is not None else [xs[min(b, n_prev - 1)]]) + extra for b in range(len(fns))])
        if len(ys) == 1 or defer:
            return ys
        return nnops.exchange(ys, self.fuse_layers, self.training, first_only=first_only)


class HRFormer(nn.Module):
    def __init__(self, in_channels: int = 3, drop_path_rate: float = 0.2, with_rpe: bool = True,
                 stage1_num_modules: int = 1, stage1_num_branches: int = 1, stage1_num_blocks: Sequence[int] = (2,),
                 stage1_num_channels: Sequence[int] = (64,),
                 stage2_num_modules: int = 1, stage2_num_branches: int = 2, stage2_num_blocks=(2, 2),
                 stage2_num_channels=(78, 156), stage2_num_heads=(2, 4), stage2_mlp_ratios=(4, 4), stage2_window_sizes=(7, 7),
                 stage3_num_modules: int = 4, stage3_num_branches: int = 3, stage3_num_blocks=(2, 2, 2),
                 stage3_num_channels=(78, 156, 312), stage3_num_heads=(2, 4, 8), stage3_mlp_ratios=(4, 4, 4),
                 stage3_window_sizes=(7, 7, 7),
                 stage4_num_modules: int = 2, stage4_num_branches: int = 4, stage4_num_blocks=(2, 2, 2, 2),
                 stage4_num_channels=(78, 156, 312, 624), stage4_num_heads=(2, 4, 8, 16), stage4_mlp_ratios=(4, 4, 4, 4),
                 stage4_window_sizes=(7, 7, 7, 7)):
        super().__init__()
        self._ctor = {k: v for k, v in locals().items() if k not in ("self", "__class__")}      # to rebuild a padded twin (models/padded.py)
        for ws in (*stage2_window_sizes, *stage3_window_sizes, *stage4_window_sizes):
            if ws != 7:
                raise ValueError("window size must be 7")
        self.drop_path_rate = drop_path_rate
        self.conv1, self.bn1 = conv(in_channels, 64, 3, 2), nn.BatchNorm2d(64)
        self.conv2, self.bn2 = conv(64, 64, 3, 2), nn.BatchNorm2d(64)
        planes = stage1_num_channels[0]
        self.layer1 = nn.ModuleList(Residual(64 if i == 0 else planes * 4, planes, True, project=(i == 0 and 64 != planes * 4))
                                    for i in range(stage1_num_blocks[0]))
        pre = [planes * 4]
        for s, (nm, ch, hd, nb, mr) in enumerate(((stage2_num_modules, stage2_num_channels, stage2_num_heads, stage2_num_blocks, stage2_mlp_ratios),
                                                  (stage3_num_modules, stage3_num_channels, stage3_num_heads, stage3_num_blocks, stage3_mlp_ratios),
                                                  (stage4_num_modules, stage4_num_channels, stage4_num_heads, stage4_num_blocks, stage4_mlp_ratios)), start=2):
            setattr(self, f"transition{s - 1}", make_transition(pre, list(ch)))
            setattr(self, f"stage{s}", nn.ModuleList(HRFormerModule(list(ch), list(hd), list(nb), list(mr), drop_path_rate, with_rpe) for _ in range(nm)))
            pre = list(ch)
        self.out_channels = stage4_num_channels[0]
        init_backbone_weights(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tw = nnops.padded_twin(self)
        if tw is not None:                       # C % 8 != 0 (HRFormer-base): run the 8-aligned twin, hand back the real channels
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
        mods = [m for s in (2, 3, 4) for m in getattr(self, f"stage{s}")]
        scales = None
        if tr and self.drop_path_rate > 0:
            scales = nnops.drop_scales(sum(m.n_draws() for m in mods), x.shape[0], self.drop_path_rate, x.device)
        d = 0
        for s in (2, 3, 4):
            trans = getattr(self, f"transition{s - 1}") if _CHAIN else None
            if not _CHAIN:
                ys = run_transition(getattr(self, f"transition{s - 1}"), ys, s, tr)
            mods_s, pre = list(getattr(self, f"stage{s}")), None
            for k, m in enumerate(mods_s):
                n = m.n_draws()
                chain = _CHAIN and k + 1 < len(mods_s)          # the exchange unit of this module runs inside the next module's tasks
                ys = m(ys, None if scales is None else scales[d:d + n], pre=pre, defer=chain, trans=trans if k == 0 else None,
                       first_only=(s == 4 and k + 1 == len(mods_s)))
                pre = m.fuse_layers if chain else None
                d += n
            # N > 1: once backward has passed this boundary the later stages' gradients are exchanged while the earlier stages still
            # run backward (only output 0 of the last stage is consumed, hrformer.py:776 / hrnet.py:441)
            nnops.backward_milestone(ys if s < 4 else ys[:1])
        return ys[0]


WindowAttention, Mlp = WindowAttentionParams, MlpParams          # the reference's class names (hrformer.py:38,117)


def hrformer_base(pretrained: bool = False, **kwargs) -> HRFormer:
    """HRFormer-Base: C=(78,156,312,624), heads (2,4,8,16) -> head_dim 39, drop_path 0.2 (hrformer.py:779-825)."""
    kwargs.setdefault("drop_path_rate", 0.2)
    return HRFormer(in_channels=3, with_rpe=True, **kwargs)


def hrformer_small(pretrained: bool = False, **kwargs) -> HRFormer:
    """HRFormer-Small: C=(32,64,128,256), heads (1,2,4,8), drop_path 0.1 (hrformer.py:828-846).
    Unlike the reference (duplicate-kwarg TypeError), `drop_path_rate=` may be overridden here."""
    kwargs.setdefault("drop_path_rate", 0.1)
    return HRFormer(in_channels=3, with_rpe=True,
                    stage2_num_channels=(32, 64), stage2_num_heads=(1, 2),
                    stage3_num_channels=(32, 64, 128), stage3_num_heads=(1, 2, 4),
                    stage4_num_channels=(32, 64, 128, 256), stage4_num_heads=(1, 2, 4, 8), **kwargs)
