"""Exchange unit (hrformer.py:420-491 == hrnet.py:157-227) as GROUPED launches.

out_i = relu(sum_j route_{j->i}(x_j)): j == i identity; j > i conv1x1 + BN, bilinearly up-sampled inside the sum; j < i a chain of (i - j)
stride-2 conv3x3 + BN layers with ReLU on all but the last.  A unit of n branches is n(n-1)/2 up routes and as many down chains: 2 .. 16
small conv + BN layers on 1 500 .. 50 000 pixels.  Launched one by one (round 3) they were ~30 % of the step's 1 288 launches, each paying
~10 us of fixed cost for 1-3 us of work, and the step time followed the launch COUNT.  Here every dependency LEVEL of the unit is one
launch per kernel family:

    forward    level l:  pk_conv2d_group (all convs whose inputs are ready) -> pk_bn_train_fwd_group;   end: pk_fuse_sum_group (all outputs)
    backward   pk_upsample_bwd_group (all up routes, the sum's ReLU mask folded in);  level l, deepest first: pk_bn_bwd_group (2 launches)
               -> pk_conv2d_group (data gradients: 1x1 and dilated 3x3 separately) -> pk_wgrad_group (slabs, reduced with everything else
               by the step's one pk_reduce_many);  end: pk_fuse_sum_group (input-gradient sums, identity route's ReLU mask folded in)

n = 4: 7 launches forward and 15 backward instead of 36 and 76.  Same kernels bodies as the per-layer path (k_igemm2 128 x 32 tile,
k_bn_act_fin, k_bn_bwd_*, k_wgrad4, k_fuse_sum, k_upsample_bwd); descriptor blocks travel by value in the kernel arguments.
Nothing here touches oracle/ and there is no PyTorch fallback: eval mode WITH autograd (a rare combination) takes the per-layer path.
"""
import os

import numpy as np
import torch

from . import _lib, nnops
from ._lib import call, stream_ptr

BF16, F32 = torch.bfloat16, torch.float32
GROUP_MAX = 12

_P = "<u8"
CONV_DT = np.dtype([("x", _P), ("w", _P), ("out", _P), ("stats", _P), ("col_scale", _P), ("bias", _P), ("res", _P),
                    ("B", "<i4"), ("Hs", "<i4"), ("Ws", "<i4"), ("Cin", "<i4"), ("Cout", "<i4"), ("ksize", "<i4"), ("stride", "<i4"),
                    ("dilated_input", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"), ("act", "<i4")], align=True)
BNF_DT = np.dtype([("raw", _P), ("stats_partial", _P), ("gamma", _P), ("beta", _P), ("running_mean", _P), ("running_var", _P),
                   ("num_batches_tracked", _P), ("residual", _P), ("y", _P), ("save_mean", _P), ("save_rstd", _P), ("relu_mask", _P), ("rows", "<i8"),
                   ("tiles", "<i4"), ("C", "<i4"), ("momentum", "<f4"), ("eps", "<f4"), ("relu", "<i4")], align=True)
BNB_DT = np.dtype([("dy", _P), ("y_act", _P), ("raw", _P), ("save_mean", _P), ("save_rstd", _P), ("gamma", _P), ("partial", _P),
                   ("dgamma", _P), ("dbeta", _P), ("dx", _P), ("dresidual", _P), ("relu_mask", _P), ("rows", "<i8"), ("C", "<i4"), ("relu", "<i4")],
                  align=True)
FUSE_DT = np.dtype([("inputs", _P, (4,)), ("in_h", "<i4", (4,)), ("in_w", "<i4", (4,)), ("n_inputs", "<i4"), ("out", _P), ("mask_y", _P),
                    ("B", "<i4"), ("H", "<i4"), ("W", "<i4"), ("C", "<i4"), ("relu", "<i4")], align=True)
UPB_DT = np.dtype([("dy", _P), ("mask_y", _P), ("dsrc", _P), ("B", "<i4"), ("H", "<i4"), ("W", "<i4"), ("Hs", "<i4"), ("Ws", "<i4"),
                   ("C", "<i4")], align=True)
WG_DT = np.dtype([("x", _P), ("grad_out", _P), ("workspace", _P), ("B", "<i4"), ("Hs", "<i4"), ("Ws", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"),
                  ("N", "<i4"), ("Cin", "<i4"), ("ksize", "<i4"), ("stride", "<i4")], align=True)
for _i, _dt in enumerate((CONV_DT, BNF_DT, BNB_DT, FUSE_DT, UPB_DT, WG_DT)):
    if _lib.lib.pk_sizeof_group_desc(_i) != _dt.itemsize:
        raise _lib.PoseKernelError(f"descriptor {_i}: host layout {_dt.itemsize} B != library {_lib.lib.pk_sizeof_group_desc(_i)} B")


def enabled() -> bool:
    return os.environ.get("POSE_GROUPED_EXCHANGE", "1") != "0"


def whole_unit() -> bool:
    """POSE_GROUPED_EXCHANGE=2: an un-chained unit (last module of a stage) runs as ONE grouped unit on the current stream instead of one
    grouped task per output on the branch streams.  (Measured: with EVERY unit whole and the modules un-chained the step is 0.2 ms slower
    than per-layer launches inside chained branch tasks -- 16.83 vs 16.62 ms: fewer launches, but all of them on the critical path.)"""
    return os.environ.get("POSE_GROUPED_EXCHANGE", "1") == "2"


def _p(t):
    return 0 if t is None else t.data_ptr()


def _launch(name, dt, rows):
    """rows: list of dicts keyed by the descriptor's fields -> ceil(len / GROUP_MAX) grouped launches."""
    for s in range(0, len(rows), GROUP_MAX):
        chunk = rows[s:s + GROUP_MAX]
        arr = np.zeros(len(chunk), dtype=dt)
        for i, r in enumerate(chunk):
            for k, v in r.items():
                arr[i][k] = v
        call(name, arr.ctypes.data, len(chunk), stream_ptr())


class _Layer:
    __slots__ = ("i", "j", "step", "last", "up", "prev", "level", "conv", "bn", "relu")


class Plan:
    """Static description of (a subset of the outputs of) one exchange unit."""

    def __init__(self, fuse, n, outs):
        self.n, self.outs = n, list(outs)
        self.layers = []
        for i in self.outs:
            for j in range(n):
                if j == i:
                    continue
                chain = [fuse[str(i)][str(j)]] if j > i else list(fuse[str(i)][str(j)])
                prev = -1
                for s, (conv, bn) in enumerate(chain):
                    l = _Layer()
                    l.i, l.j, l.step, l.last, l.up, l.prev, l.level = i, j, s, s == len(chain) - 1, j > i, prev, s
                    l.conv, l.bn, l.relu = conv, bn, (j < i and s != len(chain) - 1)
                    prev = len(self.layers)
                    self.layers.append(l)
        self.n_levels = 1 + max((l.level for l in self.layers), default=-1)

    def params(self):
        out = []
        for l in self.layers:
            out += [l.conv.weight, l.bn.weight, l.bn.bias]
        return out


def _geom(x, conv):
    ks, stride = conv.weight.shape[2], conv.stride[0]
    B, Hs, Ws, Cin, Ho, Wo = nnops._conv_geometry(x, ks, stride)
    return B, Hs, Ws, Cin, Ho, Wo, ks, stride, conv.weight.shape[0]


class _Unit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, training, *tensors):
        n, L = plan.n, plan.layers
        xs = [t.contiguous() for t in tensors[:n]]
        wc = nnops._wc()
        dev = xs[0].device
        raws, ys, means, rstds, masks = [None] * len(L), [None] * len(L), [None] * len(L), [None] * len(L), [None] * len(L)
        want_grad = training and any(t.requires_grad for t in tensors)
        for lv in range(plan.n_levels):
            idx = [k for k, l in enumerate(L) if l.level == lv]
            convs, bns = [], []
            for k in idx:
                l = L[k]
                xin = xs[l.j] if l.step == 0 else ys[l.prev]
                B, Hs, Ws, Cin, Ho, Wo, ks, stride, Cout = _geom(xin, l.conv)
                M = B * Ho * Wo
                d = dict(x=_p(xin), w=_p(wc.fwd[id(l.conv.weight)]), B=B, Hs=Hs, Ws=Ws, Cin=Cin, Cout=Cout, ksize=ks, stride=stride, Ho=Ho, Wo=Wo)
                if training:
                    raws[k] = nnops._e((B, Ho, Wo, Cout), BF16, dev)
                    ys[k] = nnops._e((B, Ho, Wo, Cout), BF16, dev)
                    tiles = _lib.lib.pk_conv_stats_tiles(M)
                    part = nnops._e((tiles, 2, Cout), F32, dev)
                    means[k], rstds[k] = nnops._e((Cout,), F32, dev), nnops._e((Cout,), F32, dev)
                    d.update(out=_p(raws[k]), stats=_p(part))
                    if l.relu and want_grad:       # the chain step's own ReLU as a bit mask for its BatchNorm backward
                        masks[k] = nnops._e((M * Cout // 8,), torch.uint8, dev)
                    bn = l.bn
                    wc.bn_eval.pop(id(bn), None)        # the kernel rewrites the running statistics in place (no version bump)
                    bns.append(dict(raw=_p(raws[k]), stats_partial=_p(part), gamma=_p(bn.weight), beta=_p(bn.bias), running_mean=_p(bn.running_mean),
                                    running_var=_p(bn.running_var), num_batches_tracked=_p(bn.num_batches_tracked), y=_p(ys[k]),
                                    save_mean=_p(means[k]), save_rstd=_p(rstds[k]), relu_mask=_p(masks[k]), rows=M, tiles=tiles, C=Cout, momentum=0.1,
                                    eps=1e-5, relu=1 if l.relu else 0))
                else:                                   # eval without autograd: BatchNorm is an affine map in the conv's epilogue
                    scale, shift = nnops._bn_eval_affine(wc, l.bn, l.bn.weight, l.bn.bias)[:2]
                    ys[k] = nnops._e((B, Ho, Wo, Cout), BF16, dev)
                    d.update(out=_p(ys[k]), col_scale=_p(scale), bias=_p(shift), act=3 if l.relu else 0)
                    convs.append((d, scale, shift))
                    continue
                convs.append((d, part, None))
            _launch("pk_conv2d_group", CONV_DT, [c[0] for c in convs])
            if bns:
                _launch("pk_bn_train_fwd_group", BNF_DT, bns)
        outs, fuse_rows = [], []
        for i in plan.outs:
            terms = []
            for j in range(n):
                if j == i:
                    terms.append(xs[j])
                else:
                    k = next(k for k, l in enumerate(L) if l.i == i and l.j == j and l.last)
                    terms.append(ys[k])
            Bq, H, W, C = xs[i].shape
            out = nnops._e((Bq, H, W, C), BF16, dev)
            outs.append(out)
            fuse_rows.append(dict(inputs=[_p(t) for t in terms] + [0] * (4 - len(terms)), in_h=[t.shape[1] for t in terms] + [0] * (4 - len(terms)),
                                  in_w=[t.shape[2] for t in terms] + [0] * (4 - len(terms)), n_inputs=len(terms), out=_p(out), B=Bq, H=H, W=W, C=C, relu=1))
        _launch("pk_fuse_sum_group", FUSE_DT, fuse_rows)
        if want_grad:
            ctx.plan = plan
            ctx.n_saved = (len(xs), len(L))
            wds = [wc.dgrad[id(l.conv.weight)] for l in L]
            ctx.save_for_backward(*xs, *raws, *ys, *means, *rstds, *outs, *wds, *masks)
            ctx.params = tensors[n:]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        plan = ctx.plan
        n, L = plan.n, plan.layers
        nl = len(L)
        sv = ctx.saved_tensors
        xs, raws, ys, means, rstds = sv[:n], sv[n:n + nl], sv[n + nl:n + 2 * nl], sv[n + 2 * nl:n + 3 * nl], sv[n + 3 * nl:n + 4 * nl]
        outs = sv[n + 4 * nl:n + 4 * nl + len(plan.outs)]
        wds = sv[n + 4 * nl + len(plan.outs):n + 5 * nl + len(plan.outs)]
        masks = sv[n + 5 * nl + len(plan.outs):]
        params = ctx.params
        dev = xs[0].device
        dout = {i: douts[q].contiguous() for q, i in enumerate(plan.outs)}
        out_of = {i: outs[q] for q, i in enumerate(plan.outs)}
        # 1. up-sampling backward of every up route (the fused sum's ReLU mask is applied to dout on the fly)
        dys = {}                                         # layer -> (dy, tensor whose sign is the ReLU mask or None)
        rows = []
        for k, l in enumerate(L):
            if l.up:
                Bq, H, W, C = out_of[l.i].shape
                d = nnops._e(tuple(ys[k].shape), BF16, dev)
                dys[k] = (d, None)
                rows.append(dict(dy=_p(dout[l.i]), mask_y=_p(out_of[l.i]), dsrc=_p(d), B=Bq, H=H, W=W, Hs=d.shape[1], Ws=d.shape[2], C=C))
            elif l.last:
                dys[k] = (dout[l.i], out_of[l.i])        # BatchNorm backward masks with the SUM's output
        if rows:
            _launch("pk_upsample_bwd_group", UPB_DT, rows)
        contrib = {j: [] for j in range(n)}
        grads = [None] * (3 * nl)
        for lv in reversed(range(plan.n_levels)):
            idx = [k for k, l in enumerate(L) if l.level == lv]
            draws, bn_rows, keep = {}, [], []
            for k in idx:
                l = L[k]
                dy, mask = dys[k]
                Bq, Ho, Wo, Cout = raws[k].shape
                M = Bq * Ho * Wo
                g_p, b_p = params[3 * k + 1], params[3 * k + 2]
                (dgamma, sg), (dbeta, sb) = nnops._sink(g_p), nnops._sink(b_p)
                if not sg:
                    grads[3 * k + 1] = dgamma
                if not sb:
                    grads[3 * k + 2] = dbeta
                nb = _lib.lib.pk_bn_bwd_group_blocks(M)
                part = nnops._e((nb, 2, Cout), F32, dev)
                draws[k] = nnops._e(tuple(raws[k].shape), BF16, dev)
                keep.append(part)
                bits = masks[k] if (mask is not None and mask is ys[k]) else None      # the step's own ReLU: its bit mask instead of y
                bn_rows.append(dict(dy=_p(dy), y_act=0 if bits is not None else _p(mask), raw=_p(raws[k]), save_mean=_p(means[k]),
                                    save_rstd=_p(rstds[k]), gamma=_p(g_p), partial=_p(part), dgamma=_p(dgamma), dbeta=_p(dbeta), dx=_p(draws[k]),
                                    relu_mask=_p(bits), rows=M, C=Cout, relu=1 if mask is not None else 0))
            _launch("pk_bn_bwd_group", BNB_DT, bn_rows)
            plain, dil, wg1, wg3, single = [], [], [], [], []
            for k in idx:
                l = L[k]
                xin = xs[l.j] if l.step == 0 else ys[l.prev]
                Bq, Hs, Ws, Cin, Ho, Wo, ks, stride, Cout = _geom(xin, l.conv)
                dx = nnops._e((Bq, Hs, Ws, Cin), BF16, dev)
                row = dict(x=_p(draws[k]), w=_p(wds[k]), out=_p(dx), B=Bq, Hs=Ho, Ws=Wo, Cin=Cout, Cout=Cin, ksize=ks, stride=1,
                           dilated_input=1 if stride == 2 else 0, Ho=Hs, Wo=Ws)
                (dil if stride == 2 else plain).append(row)
                if l.step == 0:
                    contrib[l.j].append(dx)
                else:
                    dys[l.prev] = (dx, ys[l.prev])       # the previous chain step ends in a ReLU of its own
                w_p = params[3 * k]
                dst, sw = nnops._sink(w_p)
                if sw and nnops.deferral_enabled():
                    S = _lib.lib.pk_wgrad_group_slices(Bq * Ho * Wo, Cout, Cin, ks, stride)
                    total = Cout * ks * ks * Cin
                    ws = nnops._workspace(dst, "wg", S * total)
                    (wg3 if stride == 2 else wg1).append(dict(x=_p(xin), grad_out=_p(draws[k]), workspace=_p(ws), B=Bq, Hs=Hs, Ws=Ws, Ho=Ho, Wo=Wo,
                                                              N=Cout, Cin=Cin, ksize=ks, stride=stride))
                    nnops._defer(ws.data_ptr(), dst, total, S, total, 1, Cout, ks * ks, Cin)
                else:
                    single.append((k, xin, Cout, Cin, ks, stride, (Bq, Hs, Ws, Ho, Wo), dst, sw))
            for grp in (plain, dil):
                if grp:
                    _launch("pk_conv2d_group", CONV_DT, grp)
            for grp in (wg1, wg3):
                if grp:
                    _launch("pk_wgrad_group", WG_DT, grp)
            for k, xin, Cout, Cin, ks, stride, geom, dst, sw in single:      # no gradient sink (plain autograd): the per-layer kernel + its reduce
                dw = nnops._wgrad(xin, draws[k], Cout, Cin, ks, stride, geom, out=dst, deferred=False)
                grads[3 * k] = None if sw else dw
        # 3. input gradients: identity route (masked by the sum's ReLU) + the data gradients of the routes that start at x_j
        dxs, rows = [None] * n, []
        for j in range(n):
            terms = ([dout[j]] if j in dout else []) + contrib[j]
            if not terms:
                continue
            Bq, H, W, C = xs[j].shape
            dxs[j] = nnops._e((Bq, H, W, C), BF16, dev)
            rows.append(dict(inputs=[_p(t) for t in terms] + [0] * (4 - len(terms)), in_h=[H] * len(terms) + [0] * (4 - len(terms)),
                             in_w=[W] * len(terms) + [0] * (4 - len(terms)), n_inputs=len(terms), out=_p(dxs[j]),
                             mask_y=_p(out_of[j]) if j in dout else 0, B=Bq, H=H, W=W, C=C, relu=0))
        _launch("pk_fuse_sum_group", FUSE_DT, rows)
        return (None, None, *dxs, *grads)


def unit(xs, fuse, training, outs=None):
    """Outputs `outs` (default: all) of the exchange unit `fuse` on inputs xs, grouped launches.  -> list of tensors."""
    n = len(xs)
    outs = list(range(n)) if outs is None else list(outs)
    key = (id(fuse), n, tuple(outs))
    cache = getattr(fuse, "_pk_plans", None)
    if cache is None:
        cache = {}
        object.__setattr__(fuse, "_pk_plans", cache)
    plan = cache.get(key)
    if plan is None:
        plan = cache[key] = Plan(fuse, n, outs)
    if n > 4:
        raise _lib.PoseKernelError("exchange unit with more than 4 branches")
    return list(_Unit.apply(plan, training, *xs, *plan.params()))


MAX_ROWS = 65536


def usable(training, xs=None) -> bool:
    """Grouped path: training mode, or no autograd at all (eval mode WITH autograd keeps the per-layer path) -- and only while the unit is
    launch-bound: every member runs on the 128 x 32 tile, which loses to the per-layer launcher's wide tiles / halo kernel once the
    second-highest resolution has more than MAX_ROWS pixel rows.  Measured on one box: BASELINE cfg 2 (49 152 rows) 16.07 vs 16.37 ms,
    cfg 4 (55 296 rows) 20.56 vs 21.19 ms per training step in favour of the groups; cfg 5 inference (110 592 rows) 17.78 vs 16.65 ms
    against them."""
    if not (enabled() and (training or not torch.is_grad_enabled())):
        return False
    if xs is not None and len(xs) > 1:
        t = xs[1]
        if t.shape[0] * t.shape[1] * t.shape[2] > MAX_ROWS:
            return False
    return True
