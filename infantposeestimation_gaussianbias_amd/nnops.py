"""Network op layer: every dense op of HRFormer / HRNet / the heads as hand-written HIP kernels (libposekernels.so).

Feature maps are plain contiguous (B,H,W,C) bf16 tensors (NHWC).  Each op is a `torch.autograd.Function` whose
forward and backward are explicit sequences of C-ABI calls; torch only allocates the tensors and routes gradients.

  conv_bn_act   A5/A6/H1  conv3x3|1x1 (MFMA implicit GEMM, BN statistics in the epilogue) -> BN finalize -> scale/shift(+res)(+ReLU)
  head_out      H1/H2     1x1 conv + bias (+softplus) -> fp32 NCHW maps
  attn_half     A1-A3     LN1 -> qkv GEMM (window gather) -> window attention -> proj GEMM (+residual, DropPath, window scatter)
  mlp_half      A3        LN2 -> fc1 GEMM + GELU -> fc2 GEMM (+residual, DropPath)
  fuse_sum      A4        sum of branches with fused bilinear up-sampling (+ReLU)

Channel counts must be multiples of 8 (16-byte bf16 chunks) and head_dim a multiple of 8 up to 64; models outside that
(HRFormer-base C=78 / head_dim 39, HRNet-W18) run through an 8-aligned padded twin built from the same kernels
(models/padded.py).  There is no PyTorch / CPU fallback and nothing here touches oracle/.
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import call, stream_ptr

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
ACT_DTYPE = BF16
WS = 7


def _e(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


# Gradient sink: a parameter may carry `_pk_grad_sink`, an fp32 view of the engine's flat gradient buffer.  Backward kernels
# then store that parameter's gradient THERE (plain store: every parameter is used once per step) and hand None to autograd,
# which removes ~800 `grad += g` launches per step.  Set by engine.FlatAdamW(direct_grads=True); absent = ordinary autograd.
# The "once per step" assumption is checked: a step is delimited by `begin_grad_epoch()` (FlatAdamW.zero_grad), a second
# request for the same sink inside one epoch raises (a plain store would silently drop the first gradient), and the optimiser
# zeroes the sinks nobody asked for in the epoch (`sink_written`), so a stale gradient is never re-applied.
_SINK_EPOCH = [0]


def begin_grad_epoch():
    _SINK_EPOCH[0] += 1


def sink_written(param) -> bool:
    """True when a backward kernel was handed this parameter's gradient sink since the last begin_grad_epoch()."""
    return getattr(param, "_pk_epoch", -1) == _SINK_EPOCH[0]


def grad_sink_of(param):
    """Called by backward functions only (and by the padded twin for the real parameters): a parameter whose sink was asked
    for has received a gradient this step (`_pk_used`; the padded twin extracts only those, others keep grad=None)."""
    dst = getattr(param, "_pk_grad_sink", None)
    if dst is not None:
        if getattr(param, "_pk_epoch", -1) == _SINK_EPOCH[0] and _SINK_EPOCH[0] > 0:
            raise _lib.PoseKernelError("gradient sink requested twice in one step: a parameter applied more than once per forward (or a "
                                       "second backward without zero_grad) needs accumulation, which the direct-store sinks do not do; "
                                       "build the optimiser with direct_grads=False")
        param._pk_used = True
        param._pk_epoch = _SINK_EPOCH[0]
    return dst


def _sink(param):
    """-> (destination tensor, direct?) for the gradient of `param`."""
    dst = grad_sink_of(param)
    if dst is not None:
        return dst, True
    return _e(tuple(param.shape), F32, param.device), False


def _up8(n):
    return -(-n // 8) * 8


# ================================================================================================ deferred reductions
# Parameter gradients that go straight into a gradient sink are not read by anything in backward, so their final slab
# reductions (split-M weight-gradient slabs, LayerNorm dgamma/dbeta partials, rel-pos-bias partials) are not launched where
# they are produced: the producers write into persistent per-parameter workspaces and register one row per reduction;
# `finalize_deferred()` (called by whoever consumes the sinks: engine.Trainer / FlatAdamW.step / the padded twin) reduces all
# of them with ONE pk_reduce_many launch.  ~360 launches per step leave the data-gradient chain.
_REDUCE_DTYPE = np.dtype([("part", "<i8"), ("out", "<i8"), ("stride", "<i8"), ("S", "<i4"), ("K", "<i4"), ("layout", "<i4"),
                          ("N", "<i4"), ("T", "<i4"), ("Cin", "<i4"), ("ostride", "<i4"), ("pad", "<i4")])
_PENDING = []           # rows registered since the last finalize
_TABLES = {}            # key (tuple of rows) -> device table; never evicted: a captured hipGraph may hold its pointers


def deferral_enabled():
    import os
    return os.environ.get("POSE_DEFER_REDUCE", "1") != "0"


def _workspace(sink, tag, numel):
    """Persistent fp32 workspace attached to the gradient-sink tensor it serves (lives and dies with the model's gradient
    buffer, so a captured hipGraph can keep its pointer).  `freeze_workspaces` marks them after a capture."""
    d = getattr(sink, "_pk_ws", None)
    if d is None:
        d = sink._pk_ws = {}
    ws = d.get(tag)
    if ws is None or ws.numel() < numel:
        if ws is not None and getattr(sink, "_pk_ws_frozen", False):
            raise _lib.PoseKernelError("a slab workspace would have to grow after a hipGraph was captured with it (a batch larger "
                                       "than the captured one?): capture with the largest batch or use eager mode")
        ws = d[tag] = torch.empty(numel, dtype=F32, device=sink.device)
    return ws


def freeze_workspaces(model):
    """Called after a hipGraph capture: the graph holds the workspace pointers of every gradient sink of `model`."""
    mods = [model]
    tw = getattr(model, "_pk_twin", None)
    if tw:
        mods.append(tw.twin)
    for m in mods:
        for p in m.parameters():
            sk = getattr(p, "_pk_grad_sink", None)
            if sk is not None:
                sk._pk_ws_frozen = True


def _defer(part_ptr, out, slab_stride, S, K, layout=0, N=0, T=1, Cin=1, out_stride=1, out_offset=0):
    _PENDING.append((part_ptr, out.data_ptr() + 4 * out_offset, slab_stride, S, K, layout, N, T, Cin, out_stride, 0))


def _reduce_table(key, dev):
    desc = np.array(list(key), dtype=_REDUCE_DTYPE)
    blk_desc, blk_first, nb = [], [], 0
    cols = _lib.lib.pk_reduce_many_cols()
    for i, r in enumerate(key):
        k = -(-r[4] // cols)
        blk_desc += [i] * k
        blk_first += [nb] * k
        nb += k
    return dict(desc=torch.from_numpy(desc.view(np.uint8)).to(dev), nb=nb, blk_desc=torch.tensor(blk_desc, dtype=I32, device=dev),
                blk_first=torch.tensor(blk_first, dtype=I32, device=dev))


def finalize_deferred():
    """Reduce every slab set registered since the last call (no-op when nothing is pending)."""
    if not _PENDING:
        return
    key = tuple(_PENDING)
    _PENDING.clear()
    tab = _TABLES.get(key)
    if tab is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.PoseKernelError("deferred-reduction table changed during hipGraph capture (host->device table upload is not "
                                       "capturable): run at least two eager warm-up steps before capturing")
        tab = _TABLES[key] = _reduce_table(key, torch.device("cuda", torch.cuda.current_device()))
    call("pk_reduce_many", tab["desc"], tab["blk_desc"], tab["blk_first"], tab["nb"], stream_ptr())


def _reduce_now(rows, dev):
    """Slab sums whose destinations are ordinary tensors (no gradient sink): one uncached pk_reduce_many launch."""
    tab = _reduce_table(tuple(rows), dev)
    call("pk_reduce_many", tab["desc"], tab["blk_desc"], tab["blk_first"], tab["nb"], stream_ptr())
    for t in tab.values():                      # keep the table alive until the launch has consumed it
        if torch.is_tensor(t):
            t.record_stream(torch.cuda.current_stream())


# ================================================================================================ backward milestones
# A milestone is a set of tensors at a stage boundary of the forward pass; when backward has produced the gradient of every one
# of them, all layers after the boundary have finished their backward.  engine.Trainer (N > 1 ranks) registers a callback that
# runs the postponed slab reductions registered so far and launches the all-reduce of the gradient buckets that are complete,
# so the exchange overlaps with the rest of backward.  Without a callback (single GPU, inference) milestones cost nothing.
_MILESTONE_CB = [None]


def set_milestone_callback(cb):
    _MILESTONE_CB[0] = cb


def backward_milestone(tensors):
    cb = _MILESTONE_CB[0]
    if cb is None or not torch.is_grad_enabled():
        return
    ts = [t for t in tensors if torch.is_tensor(t) and t.requires_grad]
    if not ts:
        return
    left = [len(ts)]

    def hook(_g):
        left[0] -= 1
        if left[0] == 0:
            cb()

    for t in ts:
        t.register_hook(hook)


# ================================================================================================ weight cache
_PACK_DTYPE = np.dtype([("src", "<i8"), ("dst_off", "<i8"), ("N", "<i4"), ("C", "<i4"), ("T", "<i4"), ("mode", "<i4"),
                        ("Cp", "<i4"), ("Np", "<i4"), ("dst_numel", "<i8")])


class WeightCache:
    """bf16 compute copies of every conv / linear weight of a model, refreshed by ONE pk_pack_weights launch.

    Per weight: `fwd` = [N][T][Cin_pad] (forward + wgrad layout) and `dgrad` = conv: [Cin][T flipped][N_pad], linear:
    [Cin][N_pad] (the data-gradient is the same implicit GEMM with these as its weights).  The fp32 parameters stay the
    masters in the reference's layouts; call `mark_dirty()` after anything that rewrites them behind torch's back
    (the fused optimiser kernel); in-place torch writes are caught through `_version`."""

    def __init__(self, model):
        self.entries = []          # (param, N, C, T)
        seen = set()
        for m in model.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)) and id(m.weight) not in seen:
                seen.add(id(m.weight))
                w = m.weight
                N, C = w.shape[0], w.shape[1]
                T = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
                self.entries.append((w, N, C, T))
        self.fwd, self.dgrad = {}, {}
        self.bn_eval = {}          # id(BatchNorm module) -> (key, scale, shift, mean, rstd): eval-mode constants, see _ConvBnAct
        self._sig = None
        self._dirty = True
        self.flat = None

    def mark_dirty(self):
        self._dirty = True
        self.bn_eval.clear()

    def _signature(self):
        return tuple((w.data_ptr(), w._version) for w, _, _, _ in self.entries)

    def _build(self, dev):
        rows, off = [], 0
        layout = []
        for w, N, C, T in self.entries:
            Cp, Np = _up8(C), _up8(N)
            n_f = N * T * Cp
            rows.append((w.data_ptr(), off, N, C, T, 0, Cp, Np, n_f))
            layout.append(("f", w, off, (N, T, Cp)))
            off += -(-n_f // 8) * 8
            n_d = C * T * Np
            rows.append((w.data_ptr(), off, N, C, T, 1 if T > 1 else 2, Cp, Np, n_d))
            layout.append(("d", w, off, (C, T, Np)))
            off += -(-n_d // 8) * 8
        if self.flat is None or self.flat.numel() != off or self.flat.device != dev:
            self.flat = torch.empty(off, dtype=BF16, device=dev)
        desc = np.array(rows, dtype=_PACK_DTYPE)
        blk_desc, blk_first, nb = [], [], 0
        for i, r in enumerate(rows):
            k = -(-r[8] // 1024)
            blk_desc += [i] * k
            blk_first += [nb] * k
            nb += k
        self._desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        self._blk_desc = torch.tensor(blk_desc, dtype=I32, device=dev)
        self._blk_first = torch.tensor(blk_first, dtype=I32, device=dev)
        self._nb = nb
        for kind, w, o, shape in layout:
            view = self.flat[o:o + shape[0] * shape[1] * shape[2]].view(shape)
            (self.fwd if kind == "f" else self.dgrad)[id(w)] = view

    def ensure_fresh(self):
        if not self.entries:
            return
        dev = self.entries[0][0].device
        sig = self._signature()
        ptrs = tuple(s[0] for s in sig)
        if self._sig is None or tuple(s[0] for s in self._sig) != ptrs or self.flat is None or self.flat.device != dev:
            self._build(dev)
            self._dirty = True
        if self._dirty or sig != self._sig:
            call("pk_pack_weights", self.flat, self._desc, self._blk_desc, self._blk_first, self._nb, stream_ptr())
            self._sig, self._dirty = sig, False
            self.bn_eval.clear()               # eval-mode BatchNorm constants derive from parameters that just changed


def weight_cache(model) -> WeightCache:
    wc = getattr(model, "_pk_weight_cache", None)
    if wc is None:
        wc = WeightCache(model)
        object.__setattr__(model, "_pk_weight_cache", wc)
    return wc


_CURRENT = []   # stack of active WeightCache objects (set by the model's forward)


class use_weights:
    def __init__(self, model):
        self.wc = weight_cache(model)

    def __enter__(self):
        self.wc.ensure_fresh()
        _CURRENT.append(self.wc)
        return self.wc

    def __exit__(self, *a):
        _CURRENT.pop()


def _wc():
    if not _CURRENT:
        raise _lib.PoseKernelError("nnops: no active weight cache (call the model's forward, or wrap in nnops.use_weights(model))")
    return _CURRENT[-1]


# ================================================================================================ helpers
def to_features(x_nchw_f32, cpad=8):
    """(B,Cin,H,W) fp32 NCHW -> (B,H,W,cpad) bf16 NHWC, channels >= Cin zero (16-byte pixels for the stem conv).
    A bf16 (B,H,W,cpad) tensor is taken as already packed (datasets.transforms.DeviceCropper writes the stem's layout directly)."""
    if x_nchw_f32.dtype == BF16 and x_nchw_f32.dim() == 4 and x_nchw_f32.shape[-1] == cpad:
        return x_nchw_f32.contiguous()
    x = x_nchw_f32.float().contiguous()
    B, Cin, H, W = x.shape
    y = _e((B, H, W, max(cpad, _up8(Cin))), BF16, x.device)
    call("pk_nchw_f32_to_nhwc_bf16", x, None, y, B, Cin, H, W, y.shape[-1], stream_ptr())
    return y


_WGRAD_LOG = None
if os.environ.get("POSE_LOG_WGRAD") == "1":          # profiling: the weight-gradient launch shapes of a run, printed at exit
    import atexit
    import collections
    _WGRAD_LOG = collections.Counter()

    def _dump_wgrad_log():
        print("# weight-gradient launches: count x (M, N, Cin, ksize, stride, a_map, g_map, g_scale, n_bias, slices, deferred)")
        for k, v in sorted(_WGRAD_LOG.items(), key=lambda kv: -kv[1] * kv[0][0] * (kv[0][1] + kv[0][2])):
            print(f"# {v:4d} x {k}  slab MB {k[9] * k[1] * (k[3] ** 2 * k[2] + 1) * 4 / 1e6:.1f}")
    atexit.register(_dump_wgrad_log)


def _conv_geometry(x, ksize, stride):
    B, Hs, Ws, Cin = x.shape
    pad = ksize // 2
    return B, Hs, Ws, Cin, (Hs + 2 * pad - ksize) // stride + 1, (Ws + 2 * pad - ksize) // stride + 1


def _conv_raw(x, wf, Cout, ksize, stride, stats):
    B, Hs, Ws, Cin, Ho, Wo = _conv_geometry(x, ksize, stride)
    raw = _e((B, Ho, Wo, Cout), BF16, x.device)
    part = None
    if stats:
        tiles = _lib.lib.pk_conv_stats_rows(B, Hs, Ws, Cin, Cout, ksize, stride, Ho, Wo)
        part = _e((tiles, 2, Cout), F32, x.device)
    call("pk_conv2d_nhwc", x, wf, raw, part, None, B, Hs, Ws, Cin, Cout, ksize, stride, 0, Ho, Wo, 0, 0, None, stream_ptr())
    return raw, part


def _conv_dgrad(g, wd, Cin, ksize, stride, in_hw, addend=None):
    """g (B,Ho,Wo,Cout) -> dx (B,Hs,Ws,Cin) with the flipped / transposed weight copy `wd` [Cin][T][Cout_pad]; `addend` (shape of dx)
    is added in the epilogue (the skip connection's gradient of a residual block)."""
    B, Ho, Wo, Cout = g.shape
    Hs, Ws = in_hw
    dx = _e((B, Hs, Ws, Cin), BF16, g.device)
    call("pk_conv2d_nhwc", g, wd, dx, None, None, B, Ho, Wo, Cout, Cin, ksize, 1, 1 if stride == 2 else 0, Hs, Ws, 0, 0, addend, stream_ptr())
    return dx


def _wgrad(x, g, N, Cin, ksize, stride, geom, a_map=None, g_map=None, g_scale=None, g_rps=0, M=None, oihw=True, out=None,
           dbias=None, deferred=False, win=None):
    """-> fp32 gradient (N, Cin, k, k) for conv (geom=(B,Hs,Ws,Ho,Wo)) or (N, Cin) for linear (geom=None); `out` = destination.
    `dbias` (fp32, <= N entries) additionally receives the bias gradient (column sums of the same scaled/gathered rows).
    `win=(B, H, W)`: the row maps are nnops.window_rowmap(B, H, W) (lets the library take its streaming kernel, which recomputes the map).
    `deferred=True` (only when every output is a gradient sink, i.e. nothing downstream in backward reads the result): only the
    split-M slabs are written here, their reduction is postponed to finalize_deferred()."""
    T = ksize * ksize
    if geom is None:
        B = Hs = Ws = Ho = Wo = 0
        if win is not None:        # (B, H, W) token grid whose window partition the row maps are: the streaming kernel recomputes them
            B, Hs, Ws = win
    else:
        B, Hs, Ws, Ho, Wo = geom
        M = B * Ho * Wo
    flags = (1 if a_map is not None else 0) | (2 if g_map is not None else 0) | (4 if g_scale is not None else 0)
    S = _lib.lib.pk_wgrad_slices(M, N, Cin, ksize, stride, Hs, Ws, flags)
    layout = 1 if (geom is not None and oihw) else 0
    n_bias = 0 if dbias is None else dbias.numel()
    if _WGRAD_LOG is not None:
        _WGRAD_LOG[(M, N, Cin, ksize, stride, a_map is not None, g_map is not None, g_scale is not None, n_bias, S, bool(deferred))] += 1
    if deferred and out is not None and deferral_enabled():
        # slabs into a persistent workspace, reduction postponed to finalize_deferred()
        ws = _workspace(out, "w", S * N * (T * Cin + 1))
        call("pk_wgrad_bf16", x, g, ws, None, None, n_bias, a_map, g_map, g_scale, g_rps, M, N, Cin, ksize, stride, B, Hs, Ws, Ho, Wo,
             layout, stream_ptr())
        total = N * T * Cin
        _defer(ws.data_ptr(), out, total, S, total, layout, N, T, Cin)
        if n_bias:
            _defer(ws.data_ptr() + 4 * S * total, dbias, N, S, n_bias)
        return out
    ws = _e((S * N * (T * Cin + 1),), F32, x.device)
    dw = out if out is not None else _e((N, Cin, ksize, ksize) if geom is not None else (N, Cin), F32, x.device)
    call("pk_wgrad_bf16", x, g, ws, dw, dbias, n_bias, a_map, g_map, g_scale, g_rps, M, N, Cin, ksize,
         stride, B, Hs, Ws, Ho, Wo, layout, stream_ptr())
    return dw


def _colsum(g, rows, N, rowmap=None, row_scale=None, rps=0, out=None):
    nb = _lib.lib.pk_ln_bwd_blocks(rows)
    part = _e((nb, N), F32, g.device)
    out = out if out is not None else _e((N,), F32, g.device)
    call("pk_colsum_bf16", g, rowmap, row_scale, rps, part, out, rows, N, stream_ptr())
    return out


# ================================================================================================ conv + BN (+res) (+ReLU)
class SkipGrad:
    """Hand-over of a residual block's skip gradient (hrnet.py:44-52,92-102: out = relu(bn(conv_last(..conv_first(x)..)) + x)).  The block's
    LAST conv takes the skip input detached and stores the skip's gradient here in its backward; the block's FIRST conv -- whose
    backward runs later, it is upstream on the same chain -- adds it in the epilogue of its data-gradient launch.  Autograd then sees
    ONE consumer of x and does not launch an elementwise add over two full-size gradients (5 per step in HRFormer-small's layer1, one per
    BasicBlock in HRNet)."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


def _bn_eval_affine(wc, bn, gamma, beta):
    """-> (scale, shift, mean, rstd) of an eval-mode BatchNorm: constants of the parameters and the running statistics, computed once
    (4 small ATen launches) and kept in the weight cache until one of them changes."""
    key = (gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version, bn.running_mean.data_ptr(), bn.running_mean._version,
           bn.running_var.data_ptr(), bn.running_var._version)
    ent = wc.bn_eval.get(id(bn))
    if ent is None or ent[0] != key:
        rstd = torch.rsqrt(bn.running_var + 1e-5)
        scale = gamma * rstd
        shift = torch.addcmul(beta, bn.running_mean, scale, value=-1.0)
        ent = wc.bn_eval[id(bn)] = (key, scale.float().contiguous(), shift.float().contiguous(), bn.running_mean.clone(), rstd)
    return ent[1:]


class _ConvBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, bn, stride, relu, training, skip_out=None, skip_in=None, infer=False):
        ctx.params = (weight, gamma, beta)
        ctx.skip_out, ctx.skip_in = skip_out, skip_in      # skip_out: store d(residual) there; skip_in: add what was stored to dx
        wc = _wc()
        wf, wd = wc.fwd[id(weight)], wc.dgrad[id(weight)]
        Cout, Cin_real, ksize = weight.shape[0], weight.shape[1], weight.shape[2]
        x = x.contiguous()
        B, Hs, Ws, Cin, Ho, Wo = _conv_geometry(x, ksize, stride)
        M = B * Ho * Wo
        dev = x.device
        res = None if residual is None else residual.contiguous()
        if infer and not training and os.environ.get("POSE_FUSED_EVAL_BN", "1") != "0":
            # inference: the running statistics make BatchNorm a per-channel affine map -> applied (with the residual and the ReLU) in the
            # conv's epilogue: ONE launch, the pre-normalisation tensor never exists (pk_conv2d_affine_nhwc)
            scale, shift = _bn_eval_affine(wc, bn, gamma, beta)[:2]
            y = _e((B, Ho, Wo, Cout), BF16, dev)
            call("pk_conv2d_affine_nhwc", x, wf, y, scale, shift, res, 1 if relu else 0, B, Hs, Ws, Cin, Cout, ksize, stride, Ho, Wo, stream_ptr())
            return y
        raw, part = _conv_raw(x, wf, Cout, ksize, stride, training)
        scale, shift = _e((Cout,), F32, dev), _e((Cout,), F32, dev)
        mean, rstd = _e((Cout,), F32, dev), _e((Cout,), F32, dev)
        y = _e(raw.shape, BF16, dev)
        # ReLU mask as one bit per element (a byte per 16-byte chunk): BatchNorm backward reads it instead of y (1/16 of the bytes)
        mask = _e((M * Cout // 8,), torch.uint8, dev) if (relu and not infer) else None
        if training:
            wc.bn_eval.pop(id(bn), None)        # the kernel below rewrites the running statistics in place (no version bump)
            # finalize + apply: one fused launch for small tensors, two kernels otherwise (the library decides)
            call("pk_bn_train_fwd", raw, part, part.shape[0], Cout, M, gamma, beta, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                 0.1, 1e-5, res, y, mean, rstd, scale, shift, 1 if relu else 0, mask, stream_ptr())
        else:
            # eval: scale / shift are constants of the parameters -- computed once (4 small ATen launches) and kept until the weights or
            # the statistics change (was: 4 launches per BatchNorm layer and forward, 320 per HRFormer-base inference step)
            scale, shift, mean, rstd = _bn_eval_affine(wc, bn, gamma, beta)
            call("pk_bn_act", raw, scale, shift, res, y, M, Cout, 1 if relu else 0, mask, stream_ptr())
        ctx.save_for_backward(x, raw, y if mask is None else mask, mean, rstd, gamma, wd)
        ctx.has_mask = mask is not None
        ctx.meta = (stride, relu, training, residual is not None, Cin_real, ksize, (B, Hs, Ws, Ho, Wo))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, raw, y, mean, rstd, gamma, wd = ctx.saved_tensors
        stride, relu, training, has_res, Cin_real, ksize, (B, Hs, Ws, Ho, Wo) = ctx.meta
        dev, Cout, Cin, M = dy.device, raw.shape[-1], x.shape[-1], B * Ho * Wo
        dy = dy.contiguous()
        nb = _lib.lib.pk_bn_bwd_blocks(M)
        w_p, g_p, b_p = ctx.params
        part, sums = _e((nb, 2, Cout), F32, dev), _e((2 * Cout,), F32, dev)
        (dgamma, sg), (dbeta, sb) = _sink(g_p), _sink(b_p)
        draw = _e(raw.shape, BF16, dev)
        dres = _e(raw.shape, BF16, dev) if has_res else None
        # (eval mode, bit 1: the running statistics are constants, the input gradient is gamma * rstd * g without the batch-mean terms)
        call("pk_bn_bwd", dy, None if ctx.has_mask else y, raw, mean, rstd, gamma, part, sums, dgamma, dbeta, draw, dres, M, Cout,
             (1 if relu else 0) | (0 if training else 2), y if ctx.has_mask else None, stream_ptr())
        addend = None
        if ctx.skip_in is not None:
            addend, ctx.skip_in.g = ctx.skip_in.g, None
            if addend is None:
                raise _lib.PoseKernelError("residual block backward out of order: the skip gradient has not been produced yet")
        if ctx.skip_out is not None:
            ctx.skip_out.g, dres = dres, None                # handed to the block's first conv instead of to autograd
        dx = _conv_dgrad(draw, wd, Cin, ksize, stride, (Hs, Ws), addend) if (ctx.needs_input_grad[0] or addend is not None) else None
        if Cin != Cin_real:                # stem: 3 real input channels inside 8-channel pixels
            dw = _wgrad(x, draw, Cout, Cin, ksize, stride, (B, Hs, Ws, Ho, Wo))[:, :Cin_real].contiguous()
            dst = grad_sink_of(w_p)
            if dst is not None:
                dst.copy_(dw)
                dw = None
        else:
            dst, sw = _sink(w_p)
            dw = _wgrad(x, draw, Cout, Cin, ksize, stride, (B, Hs, Ws, Ho, Wo), out=dst, deferred=sw)
            dw = None if sw else dw
        return dx, dw, None if sg else dgamma, None if sb else dbeta, dres, None, None, None, None, None, None, None


def conv_bn_act(x, conv, bn, relu=False, residual=None, training=False, skip_out=None, skip_in=None):
    # (grad mode is always off INSIDE an autograd Function's forward: whether nothing will be differentiated is decided here)
    return _ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, conv.stride[0], relu, training, skip_out, skip_in,
                            not torch.is_grad_enabled())


def residual_block(x, first, middle, last, training):
    """relu(bn(conv_last(...relu(bn(conv_first(x)))...)) + x) with an IDENTITY skip; first / last = (conv, bn), middle = list of (conv, bn).
    In training the skip gradient bypasses autograd (SkipGrad); otherwise this is the plain chain."""
    fuse = training and torch.is_grad_enabled() and x.requires_grad and first[0].stride[0] == 1 and os.environ.get("POSE_SKIP_GRAD", "1") != "0"
    hold = SkipGrad() if fuse else None
    y = conv_bn_act(x, first[0], first[1], True, None, training, skip_in=hold)
    for conv, bn in middle:
        y = conv_bn_act(y, conv, bn, True, None, training)
    return conv_bn_act(y, last[0], last[1], True, x.detach() if fuse else x, training, skip_out=hold)


# ================================================================================================ transposed conv (deconv head)
# HeatmapHead's optional SimpleBaseline-style stack (pose_estimator.py:47-69): ConvTranspose2d(k, stride 2) + BN + ReLU with k = 4
# (padding 1) or k = 2 (padding 0); the reference's own padding rule gives output_padding -1 for k = 3 and raises there.  A stride-2
# transposed convolution is four ordinary convolutions, one per output parity class (oy & 1, ox & 1), each using the taps ky with
# (py + p - ky) even at the input offset dy = (py + p - ky) / 2 in {-1, 0, 1}: so it runs as ONE 3x3 stride-1 convolution Cin -> 4 Cout with
# the class weights stacked on the output channels (k_igemm2 / the halo kernel, taps a class does not use are zero), followed by a pixel
# shuffle, which in NHWC is a row permutation (pk_rows_by_map).  The reference never enables the stack (num_deconv_layers = 0 at its only
# call site): correctness over speed -- 9/4 (k = 4) of the minimal MFMA work.
_SHUFFLE_MAPS = {}


def _deconv_taps(k):
    """-> list of (class = py*2+px, tap = (dy+1)*3+(dx+1), ky, kx) of a stride-2 ConvTranspose2d with kernel k, padding (k-1)//2."""
    p = (k - 1) // 2
    if k - 2 * p - 2 < 0:
        raise ValueError(f"ConvTranspose2d kernel {k}: output_padding {k - 2 * p - 2} (the reference's rule, pose_estimator.py:53-54) is negative")
    one = []
    for par in (0, 1):
        for kk in range(k):
            if (par + p - kk) % 2 == 0 and -1 <= (par + p - kk) // 2 <= 1:
                one.append((par, (par + p - kk) // 2, kk))
    return [(py * 2 + px, (dy + 1) * 3 + (dx + 1), ky, kx) for py, dy, ky in one for px, dx, kx in one]


class _StackDeconv(torch.autograd.Function):
    """ConvTranspose2d weight (Cin, Cout, k, k) -> the stacked 3x3 conv weight (4 Cout, Cin, 3, 3) of the parity-class form; backward maps
    the stacked gradient back (into the parameter's gradient sink when it has one)."""

    @staticmethod
    def forward(ctx, w):
        Cin, Cout, k, _ = w.shape
        ws = torch.zeros(4 * Cout, Cin, 3, 3, dtype=F32, device=w.device)
        for cls, tap, ky, kx in _deconv_taps(k):
            ws[cls * Cout:(cls + 1) * Cout, :, tap // 3, tap % 3] = w[:, :, ky, kx].t()
        ctx.param, ctx.shape = w, tuple(w.shape)
        return ws

    @staticmethod
    def backward(ctx, g):
        Cin, Cout, k, _ = ctx.shape
        dst, direct = _sink(ctx.param)
        dst.zero_()
        for cls, tap, ky, kx in _deconv_taps(k):
            dst[:, :, ky, kx] = g[cls * Cout:(cls + 1) * Cout, :, tap // 3, tap % 3].t()
        return None if direct else dst


class _ConvW(torch.autograd.Function):
    """Stride-1 convolution (3x3 / 1x1) with an EXPLICIT fp32 OIHW weight tensor (not a parameter of the weight cache): the bf16 compute
    copies are made here.  -> raw bf16 (B,H,W,N) and the BatchNorm partial statistics of the conv epilogue."""

    @staticmethod
    def forward(ctx, x, w):
        N, Cin, k, _ = w.shape
        wf = w.permute(0, 2, 3, 1).reshape(N, k * k, Cin).to(BF16).contiguous()                 # [N][T][Cin]
        wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, k * k, N).to(BF16).contiguous()      # [Cin][T flipped][N]
        x = x.contiguous()
        raw, part = _conv_raw(x, wf, N, k, 1, True)
        ctx.save_for_backward(x, wd)
        ctx.meta = (N, Cin, k)
        ctx.mark_non_differentiable(part)
        return raw, part

    @staticmethod
    def backward(ctx, g, _gpart):
        x, wd = ctx.saved_tensors
        N, Cin, k = ctx.meta
        B, H, W, _ = x.shape
        g = g.contiguous()
        dx = _conv_dgrad(g, wd, Cin, k, 1, (H, W)) if ctx.needs_input_grad[0] else None
        dw = _wgrad(x, g, N, Cin, k, 1, (B, H, W, H, W))
        return dx, dw


class _BnActOnly(torch.autograd.Function):
    """BatchNorm (+ReLU) of a raw conv output whose partial statistics [rows][2][C] are handed in (train mode) / on running statistics."""

    @staticmethod
    def forward(ctx, raw, part, gamma, beta, bn, relu, training):
        ctx.params = (gamma, beta)
        raw = raw.contiguous()
        C = raw.shape[-1]
        M = raw.numel() // C
        dev = raw.device
        y = _e(tuple(raw.shape), BF16, dev)
        if training:
            _wc().bn_eval.pop(id(bn), None)
            scale, shift, mean, rstd = (_e((C,), F32, dev) for _ in range(4))
            call("pk_bn_train_fwd", raw, part, part.shape[0], C, M, gamma, beta, bn.running_mean, bn.running_var, bn.num_batches_tracked, 0.1, 1e-5,
                 None, y, mean, rstd, scale, shift, 1 if relu else 0, None, stream_ptr())
        else:
            scale, shift, mean, rstd = _bn_eval_affine(_wc(), bn, gamma, beta)
            call("pk_bn_act", raw, scale, shift, None, y, M, C, 1 if relu else 0, None, stream_ptr())
        ctx.save_for_backward(raw, y, mean, rstd, gamma)
        ctx.meta = (relu, training, M, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        raw, y, mean, rstd, gamma = ctx.saved_tensors
        relu, training, M, C = ctx.meta
        dev = dy.device
        g_p, b_p = ctx.params
        (dgamma, sg), (dbeta, sb) = _sink(g_p), _sink(b_p)
        nb = _lib.lib.pk_bn_bwd_blocks(M)
        part, sums = _e((nb, 2, C), F32, dev), _e((2 * C,), F32, dev)
        draw = _e(tuple(raw.shape), BF16, dev)
        call("pk_bn_bwd", dy.contiguous(), y, raw, mean, rstd, gamma, part, sums, dgamma, dbeta, draw, None, M, C, (1 if relu else 0) | (0 if training else 2),
             None, stream_ptr())
        return draw, None, None if sg else dgamma, None if sb else dbeta, None, None, None


def deconv_bn_relu(x, deconv, bn, training):
    """(B,H,W,Cin) -> (B,2H,2W,Cout): ConvTranspose2d(k in {2, 4}, stride 2, bias=False) + BatchNorm + ReLU (pose_estimator.py:47-69)."""
    from . import hipops
    w = deconv.weight
    Cin, Cout, k, _ = w.shape
    if deconv.stride[0] != 2 or deconv.bias is not None or Cout % 8 or Cin % 8:
        raise _lib.PoseKernelError("deconv layer: stride-2 ConvTranspose2d without bias and channel counts that are multiples of 8")
    B, H, W, _ = x.shape
    raw_s, part = _ConvW.apply(x, _StackDeconv.apply(w))                  # (B,H,W,4*Cout): class-major channels
    key = (B, H, W, str(x.device))
    if key not in _SHUFFLE_MAPS:          # row of out pixel (b, 2i+py, 2j+px) <- row ((b*H+i)*W+j)*4 + py*2+px of the (.., Cout) row view
        b, oy, ox = torch.meshgrid(torch.arange(B), torch.arange(2 * H), torch.arange(2 * W), indexing="ij")
        m = (((b * H + oy // 2) * W + ox // 2) * 4 + (oy % 2) * 2 + (ox % 2)).reshape(-1)
        _SHUFFLE_MAPS[key] = m.to(torch.int32).to(x.device)
    raw = hipops.rows_by_map(raw_s.view(B * H * W * 4, Cout), _SHUFFLE_MAPS[key], B * 4 * H * W, False).view(B, 2 * H, 2 * W, Cout)
    t = part.shape[0]
    part4 = part.view(t, 2, 4, Cout).permute(0, 2, 1, 3).reshape(4 * t, 2, Cout).contiguous()      # a channel's statistics: its four classes
    return _BnActOnly.apply(raw, part4, bn.weight, bn.bias, bn, True, training)


# ================================================================================================ head output conv
class _HeadOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, softplus):
        ctx.params = (weight, bias)
        wc = _wc()
        wf, wd = wc.fwd[id(weight)], wc.dgrad[id(weight)]
        x = x.contiguous()
        B, H, W, Cin = x.shape
        N = weight.shape[0]
        out = _e((B, N, H, W), F32, x.device)
        call("pk_conv2d_nhwc", x, wf, out, None, bias, B, H, W, Cin, N, 1, 1, 0, H, W, 2 if softplus else 0, 2, None, stream_ptr())
        ctx.save_for_backward(x, out if softplus else x.new_empty(0), wd)
        ctx.meta = (softplus, N)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, wd = ctx.saved_tensors
        softplus, N = ctx.meta
        B, H, W, Cin = x.shape
        Np = _up8(N)
        g = _e((B, H, W, Np), BF16, x.device)
        call("pk_nchw_f32_to_nhwc_bf16", dout.contiguous(), y if softplus else None, g, B, N, H, W, Np, stream_ptr())
        dx = _e((B, H, W, Cin), BF16, x.device)
        call("pk_conv2d_nhwc", g, wd, dx, None, None, B, H, W, Np, Cin, 1, 1, 0, H, W, 0, 0, None, stream_ptr())
        w_p, b_p = ctx.params
        dst_w, dst_b = grad_sink_of(w_p), grad_sink_of(b_p)
        if dst_w is not None and dst_b is not None and deferral_enabled():
            # slabs only; the step's one pk_reduce_many sums the first N of the Np (8-aligned) rows straight into the sinks (was: a reduce
            # launch + two copies per head output, at the very start of backward)
            M = B * H * W
            S = _lib.lib.pk_wgrad_slices(M, Np, Cin, 1, 1, H, W, 0)
            ws = _workspace(dst_w, "w", S * Np * (Cin + 1))
            call("pk_wgrad_bf16", x, g, ws, None, None, N, None, None, None, 0, M, Np, Cin, 1, 1, B, H, W, H, W, 1, stream_ptr())
            _defer(ws.data_ptr(), dst_w, Np * Cin, S, N * Cin)
            _defer(ws.data_ptr() + 4 * S * Np * Cin, dst_b, Np, S, N)
            return dx, None, None, None
        db = _e((N,), F32, x.device)
        dw = _wgrad(x, g, Np, Cin, 1, 1, (B, H, W, H, W), dbias=db)[:N]
        if dst_w is not None and dst_b is not None:
            dst_w.copy_(dw)
            dst_b.copy_(db)
            return dx, None, None, None
        return dx, dw.contiguous(), db.contiguous(), None


def head_out(x, conv, softplus=False):
    return _HeadOut.apply(x, conv.weight, conv.bias, softplus)


# ================================================================================================ HRFormer block halves
_MAPS = {}


def window_rowmap(B, H, W, device):
    """int32 map: window-order token (b, wy, wx, ty, tx) -> pixel row b*H*W + y*W + x, or -1 for the zero-pad tokens
    appended at the bottom/right (hrformer.py:80-89).  Built once per shape."""
    key = (B, H, W, str(device))
    if key not in _MAPS:
        nh, nw = -(-H // WS), -(-W // WS)
        ys = (torch.arange(nh)[:, None] * WS + torch.arange(WS)[None, :])            # (nh, ws)
        xs = (torch.arange(nw)[:, None] * WS + torch.arange(WS)[None, :])            # (nw, ws)
        yy = ys[:, None, :, None].expand(nh, nw, WS, WS)
        xx = xs[None, :, None, :].expand(nh, nw, WS, WS)
        pix = torch.where((yy < H) & (xx < W), yy * W + xx, torch.full_like(yy, -1)).reshape(-1)
        full = pix[None, :].repeat(B, 1)
        full = torch.where(full >= 0, full + torch.arange(B)[:, None] * (H * W), full)
        _MAPS[key] = (full.reshape(-1).to(torch.int32).to(device), nh * nw)
    return _MAPS[key]


def _layernorm(x2d, gamma, beta, c_real=0):
    """c_real: number of real channels when the rows are zero-padded (padded twin of HRFormer-base), 0 = all."""
    M, C = x2d.shape
    y = _e((M, C), BF16, x2d.device)
    mean, rstd = _e((M,), F32, x2d.device), _e((M,), F32, x2d.device)
    call("pk_layernorm_fwd", x2d, gamma, beta, y, mean, rstd, M, C, c_real, 1e-5, stream_ptr())
    return y, mean, rstd


def _layernorm_bwd(dy, x2d, mean, rstd, gamma, dres, g_param=None, b_param=None, c_real=0):
    M, C = x2d.shape
    nb = _lib.lib.pk_ln_bwd_blocks(M)
    dx = _e((M, C), BF16, x2d.device)
    (dg, sg) = _sink(g_param) if g_param is not None else (_e((C,), F32, x2d.device), False)
    (db, sb) = _sink(b_param) if b_param is not None else (_e((C,), F32, x2d.device), False)
    if sg and sb and deferral_enabled():
        part = _workspace(dg, "ln", nb * 2 * C)
        call("pk_layernorm_bwd", dy, x2d, mean, rstd, gamma, dres, dx, part, None, None, M, C, c_real, stream_ptr())
        _defer(part.data_ptr(), dg, 2 * C, nb, C)
        _defer(part.data_ptr() + 4 * C, db, 2 * C, nb, C)
        return dx, None, None
    part = _e((nb, 2, C), F32, x2d.device)
    call("pk_layernorm_bwd", dy, x2d, mean, rstd, gamma, dres, dx, part, dg, db, M, C, c_real, stream_ptr())
    return dx, (None if sg else dg), (None if sb else db)


def _linear(x, w, out_rows, N, K, bias=None, residual=None, res_scale=None, a_map=None, o_map=None, preact=None, gelu_of=None,
            M=None, rps=0, act=0):
    out = _e((out_rows, N), BF16, x.device)
    call("pk_linear_bf16", x, w, out, bias, residual, res_scale, a_map, o_map, preact, gelu_of, M if M is not None else out_rows, N, K,
         rps, act, 0, stream_ptr())
    return out


class _AttnHalf(torch.autograd.Function):
    """x + s1 * proj(window_attention(qkv(LN1(x))))   on (B,H,W,C) bf16."""

    @staticmethod
    def forward(ctx, x, g1, b1, table, wqkv, bqkv, wproj, bproj, scale1, heads, c_real=0, attn_scale=0.0):
        ctx.params = (g1, b1, table, wqkv, bqkv, wproj, bproj)
        ctx.pad = (c_real, attn_scale)
        wc = _wc()
        x = x.contiguous()
        B, H, W, C = x.shape
        M = B * H * W
        amap, nwin = window_rowmap(B, H, W, x.device)
        Mw = B * nwin * WS * WS
        x2 = x.view(M, C)
        u, mean, rstd = _layernorm(x2, g1, b1, c_real)
        qkv = _linear(u, wc.fwd[id(wqkv)], Mw, 3 * C, C, bias=bqkv, a_map=amap)
        o = _e((Mw, C), BF16, x.device)
        lse = _e((B * nwin * heads * WS * WS,), F32, x.device)
        call("pk_window_attn_fwd", qkv, table, o, lse, B * nwin, heads, C, attn_scale, stream_ptr())
        s1 = None if scale1 is None else scale1.float().contiguous()
        y = _linear(o, wc.fwd[id(wproj)], M, C, C, bias=bproj, residual=x2, res_scale=s1, o_map=amap, M=Mw, rps=H * W)
        ctx.save_for_backward(x2, u, mean, rstd, qkv, o, lse, g1, table, s1 if s1 is not None else x.new_empty(0),
                              wc.dgrad[id(wqkv)], wc.dgrad[id(wproj)], amap)
        ctx.meta = (B, H, W, C, heads, nwin, s1 is not None)
        return y.view(B, H, W, C)

    @staticmethod
    def backward(ctx, dy):
        x2, u, mean, rstd, qkv, o, lse, g1, table, s1, wqkv_t, wproj_t, amap = ctx.saved_tensors
        B, H, W, C, heads, nwin, has_s = ctx.meta
        s1 = s1 if has_s else None
        M, Mw = B * H * W, B * nwin * WS * WS
        dev = dy.device
        dy2 = dy.contiguous().view(M, C)
        # proj: rows of the GEMM are window-order tokens, dy rows are gathered through the map; DropPath scale per sample
        pg1, pb1, ptab, pwqkv, pbqkv, pwproj, pbproj = ctx.params
        d_o = _linear(dy2, wproj_t, Mw, C, C, res_scale=s1, a_map=amap, rps=nwin * WS * WS)
        dst, s_wp = _sink(pwproj)
        dbproj, s_bp = _sink(pbproj)
        dwproj = _wgrad(o, dy2, C, C, 1, 1, None, g_map=amap, g_scale=s1, g_rps=H * W, M=Mw, out=dst, dbias=dbproj,
                        deferred=s_wp and s_bp, win=(B, H, W))
        # attention core
        dqkv = _e((Mw, 3 * C), BF16, dev)
        dtable, s_t = _sink(ptab)
        n_part = _lib.lib.pk_window_attn_bwd_ws_floats(B * nwin, heads)
        if s_t and deferral_enabled():
            part = _workspace(dtable, "rpb", n_part)
            call("pk_window_attn_bwd", qkv, table, o, d_o, lse, dqkv, part, None, B * nwin, heads, C, ctx.pad[1], stream_ptr())
            per_head = n_part // 169 // heads
            for hh in range(heads):      # group g = k*heads + h holds 169 partial sums of head h
                _defer(part.data_ptr() + 4 * 169 * hh, dtable, 169 * heads, per_head, 169, out_stride=heads, out_offset=hh)
        else:
            part = _e((n_part,), F32, dev)
            call("pk_window_attn_bwd", qkv, table, o, d_o, lse, dqkv, part, dtable, B * nwin, heads, C, ctx.pad[1], stream_ptr())
        # qkv linear: scatter the token gradients back to pixel rows (pad tokens dropped)
        du = _linear(dqkv, wqkv_t, M, C, 3 * C, o_map=amap, M=Mw)
        dst, s_wq = _sink(pwqkv)
        dbqkv, s_bq = _sink(pbqkv)
        dwqkv = _wgrad(u, dqkv, 3 * C, C, 1, 1, None, a_map=amap, M=Mw, out=dst, dbias=dbqkv, deferred=s_wq and s_bq, win=(B, H, W))
        dx, dg1, db1 = _layernorm_bwd(du, x2, mean, rstd, g1, dy2, pg1, pb1, ctx.pad[0])
        return (dx.view(B, H, W, C), dg1, db1, None if s_t else dtable, None if s_wq else dwqkv, None if s_bq else dbqkv,
                None if s_wp else dwproj, None if s_bp else dbproj, None, None, None, None)


class _MlpHalf(torch.autograd.Function):
    """x + s2 * fc2(gelu(fc1(LN2(x))))   on (B,H,W,C) bf16."""

    @staticmethod
    def forward(ctx, x, g2, b2, w1, bias1, w2, bias2, scale2, c_real=0):
        ctx.params = (g2, b2, w1, bias1, w2, bias2)
        ctx.c_real = c_real
        wc = _wc()
        x = x.contiguous()
        B, H, W, C = x.shape
        M, Hd = B * H * W, w1.shape[0]
        x2 = x.view(M, C)
        v, mean, rstd = _layernorm(x2, g2, b2, c_real)
        z = _e((M, Hd), BF16, x.device)
        h = _linear(v, wc.fwd[id(w1)], M, Hd, C, bias=bias1, preact=z, act=1)
        s2 = None if scale2 is None else scale2.float().contiguous()
        y = _linear(h, wc.fwd[id(w2)], M, C, Hd, bias=bias2, residual=x2, res_scale=s2, rps=H * W)
        ctx.save_for_backward(x2, v, mean, rstd, z, h, g2, s2 if s2 is not None else x.new_empty(0), wc.dgrad[id(w1)], wc.dgrad[id(w2)])
        ctx.meta = (B, H, W, C, Hd, s2 is not None)
        return y.view(B, H, W, C)

    @staticmethod
    def backward(ctx, dy):
        x2, v, mean, rstd, z, h, g2, s2, w1_t, w2_t = ctx.saved_tensors
        B, H, W, C, Hd, has_s = ctx.meta
        s2 = s2 if has_s else None
        M = B * H * W
        dy2 = dy.contiguous().view(M, C)
        pg2, pb2, pw1, pbias1, pw2, pbias2 = ctx.params
        dz = _linear(dy2, w2_t, M, Hd, C, res_scale=s2, gelu_of=z, rps=H * W)        # (dy W2) * s2 * gelu'(z)
        dst, s_w2 = _sink(pw2)
        db2, s_b2 = _sink(pbias2)
        dw2 = _wgrad(h, dy2, C, Hd, 1, 1, None, g_scale=s2, g_rps=H * W, M=M, out=dst, dbias=db2, deferred=s_w2 and s_b2)
        dv = _linear(dz, w1_t, M, C, Hd)
        dst, s_w1 = _sink(pw1)
        db1, s_b1 = _sink(pbias1)
        dw1 = _wgrad(v, dz, Hd, C, 1, 1, None, M=M, out=dst, dbias=db1, deferred=s_w1 and s_b1)
        dx, dg2, dbt2 = _layernorm_bwd(dv, x2, mean, rstd, g2, dy2, pg2, pb2, ctx.c_real)
        return (dx.view(B, H, W, C), dg2, dbt2, None if s_w1 else dw1, None if s_b1 else db1, None if s_w2 else dw2,
                None if s_b2 else db2, None, None)


def _fused_widths(var, default):
    import os
    return tuple(int(v) for v in os.environ.get(var, default).split(",") if v.strip())


class _WindowAttnOnly(torch.autograd.Function):
    """proj(window_attention(qkv(tokens))) on window-ordered tokens (B_*49, C): the reference's WindowAttention.forward by itself
    (hrformer.py:174-200), without the block's LayerNorm / window partition / residual."""

    @staticmethod
    def forward(ctx, tok, table, wqkv, bqkv, wproj, bproj, heads):
        ctx.params = (table, wqkv, bqkv, wproj, bproj)
        wc = _wc()
        tok = tok.contiguous()
        Mw, C = tok.shape
        nw = Mw // (WS * WS)
        qkv = _linear(tok, wc.fwd[id(wqkv)], Mw, 3 * C, C, bias=bqkv)
        o = _e((Mw, C), BF16, tok.device)
        lse = _e((nw * heads * WS * WS,), F32, tok.device)
        call("pk_window_attn_fwd", qkv, table, o, lse, nw, heads, C, 0.0, stream_ptr())
        y = _linear(o, wc.fwd[id(wproj)], Mw, C, C, bias=bproj)
        ctx.save_for_backward(tok, qkv, o, lse, table, wc.dgrad[id(wqkv)], wc.dgrad[id(wproj)])
        ctx.heads = heads
        return y

    @staticmethod
    def backward(ctx, dy):
        tok, qkv, o, lse, table, wqkv_t, wproj_t = ctx.saved_tensors
        heads, (Mw, C), dev = ctx.heads, tok.shape, dy.device
        nw = Mw // (WS * WS)
        dy = dy.contiguous()
        ptab, pwqkv, pbqkv, pwproj, pbproj = ctx.params
        d_o = _linear(dy, wproj_t, Mw, C, C)
        dbproj, dwproj = _e((C,), F32, dev), None
        dwproj = _wgrad(o, dy, C, C, 1, 1, None, M=Mw, dbias=dbproj)
        dqkv, dtable = _e((Mw, 3 * C), BF16, dev), _e(tuple(table.shape), F32, dev)
        part = _e((_lib.lib.pk_window_attn_bwd_ws_floats(nw, heads),), F32, dev)
        call("pk_window_attn_bwd", qkv, table, o, d_o, lse, dqkv, part, dtable, nw, heads, C, 0.0, stream_ptr())
        dtok = _linear(dqkv, wqkv_t, Mw, C, 3 * C)
        dbqkv = _e((3 * C,), F32, dev)
        dwqkv = _wgrad(tok, dqkv, 3 * C, C, 1, 1, None, M=Mw, dbias=dbqkv)
        return dtok, dtable, dwqkv, dbqkv, dwproj, dbproj, None


class _MlpOnly(torch.autograd.Function):
    """fc2(gelu(fc1(x))) on rows (M, C): the reference's Mlp.forward by itself (hrformer.py:38-64)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        wc = _wc()
        x = x.contiguous()
        M, C = x.shape
        Hd = w1.shape[0]
        z = _e((M, Hd), BF16, x.device)
        h = _linear(x, wc.fwd[id(w1)], M, Hd, C, bias=b1, preact=z, act=1)
        y = _linear(h, wc.fwd[id(w2)], M, w2.shape[0], Hd, bias=b2)
        ctx.save_for_backward(x, z, h, wc.dgrad[id(w1)], wc.dgrad[id(w2)])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, h, w1_t, w2_t = ctx.saved_tensors
        (M, C), Hd, N, dev = x.shape, z.shape[1], dy.shape[1], dy.device
        dy = dy.contiguous()
        dz = _linear(dy, w2_t, M, Hd, N, gelu_of=z)
        db2, db1 = _e((N,), F32, dev), _e((Hd,), F32, dev)
        dw2 = _wgrad(h, dy, N, Hd, 1, 1, None, M=M, dbias=db2)
        dx = _linear(dz, w1_t, M, C, Hd)
        dw1 = _wgrad(x, dz, Hd, C, 1, 1, None, M=M, dbias=db1)
        return dx, dw1, db1, dw2, db2


_ZERO_TABLES = {}


def rel_table(attn, heads):
    """The (169, heads) relative-position-bias table of a WindowAttention module; `with_rpe=False` modules (hrformer.py:145-191) have
    none: the kernels then add a constant all-zero table (its gradient is computed and dropped)."""
    t = getattr(attn, "relative_position_bias_table", None)
    if t is not None:
        return t
    dev = attn.qkv.weight.device
    key = (str(dev), heads)
    if key not in _ZERO_TABLES:
        _ZERO_TABLES[key] = torch.zeros(169, heads, dtype=F32, device=dev)
        _ZERO_TABLES[key]._pk_zero_table = True
    return _ZERO_TABLES[key]


def window_attention_tokens(tok, attn, heads):
    return _WindowAttnOnly.apply(tok, rel_table(attn, heads), attn.qkv.weight, attn.qkv.bias, attn.proj.weight, attn.proj.bias, heads)


def mlp_rows(x2d, mlp):
    return _MlpOnly.apply(x2d, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias)


def fused_mlp_enabled(C, c_real=0):
    """POSE_FUSED_MLP = comma-separated channel counts that take the fused MLP half (default: all the kernels are built for)."""
    return c_real in (0, C) and C in _fused_widths("POSE_FUSED_MLP", "32,64") and bool(_lib.lib.pk_ln_mlp_supported(C))


class _MlpHalfFused(torch.autograd.Function):
    """x + s2 * fc2(gelu(fc1(LN2(x)))) as ONE launch (pk_ln_mlp_fwd, C = 32 / 64): the 4C hidden never reaches HBM and nothing but
    x is saved; backward = pk_ln_mlp_bwd_dx (dx + LayerNorm partials) and pk_ln_mlp_bwd_dw (weight / bias slabs), both
    recomputing the hidden from x."""

    @staticmethod
    def forward(ctx, x, g2, b2, w1, bias1, w2, bias2, scale2):
        ctx.params = (g2, b2, w1, bias1, w2, bias2)
        wc = _wc()
        x = x.contiguous()
        B, H, W, C = x.shape
        M = B * H * W
        s2 = None if scale2 is None else scale2.float().contiguous()
        y = _e((B, H, W, C), BF16, x.device)
        call("pk_ln_mlp_fwd", x, g2, b2, wc.fwd[id(w1)], bias1, wc.fwd[id(w2)], bias2, s2, y, M, C, H * W, 1e-5, stream_ptr())
        ctx.save_for_backward(x, g2, b2, bias1, s2 if s2 is not None else x.new_empty(0), wc.fwd[id(w1)], wc.dgrad[id(w1)], wc.dgrad[id(w2)])
        ctx.meta = (B, H, W, C, s2 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g2, b2, bias1, s2, w1f, w1t, w2t = ctx.saved_tensors
        B, H, W, C, has_s = ctx.meta
        s2 = s2 if has_s else None
        M, Hd, dev = B * H * W, 4 * C, dy.device
        dy = dy.contiguous()
        pg2, pb2, pw1, pbias1, pw2, pbias2 = ctx.params
        (dg, sg), (db, sb) = _sink(pg2), _sink(pb2)
        (dw1, s_w1), (db1, s_b1), (dw2, s_w2), (dbb2, s_b2) = _sink(pw1), _sink(pbias1), _sink(pw2), _sink(pbias2)
        all_sinks = sg and sb and s_w1 and s_b1 and s_w2 and s_b2 and deferral_enabled()
        nbx, nbw = _lib.lib.pk_ln_mlp_dx_blocks(M, C), _lib.lib.pk_ln_mlp_dw_blocks(M, C)
        HS, SL = _lib.lib.pk_ln_mlp_hidden_slice(C), _lib.lib.pk_ln_mlp_slab_floats(C)
        ns = Hd // HS
        if all_sinks:
            lnp, slabs = _workspace(dg, "mlp_ln", nbx * 2 * C), _workspace(dw1, "mlp_w", ns * nbw * SL)
        else:
            lnp, slabs = _e((nbx * 2 * C,), F32, dev), _e((ns * nbw * SL,), F32, dev)
        dx = _e((B, H, W, C), BF16, dev)
        call("pk_ln_mlp_bwd_dx", dy, x, g2, b2, w1f, bias1, w1t, w2t, s2, dx, lnp, M, C, H * W, 1e-5, stream_ptr())
        call("pk_ln_mlp_bwd_dw", dy, x, g2, b2, w1f, bias1, w2t, s2, slabs, M, C, H * W, 1e-5, stream_ptr())
        rows = []

        def red(part_off, out, S, K, stride, layout=0, T=1, Cin=1, out_stride=1, out_offset=0, base=slabs):
            rows.append((base.data_ptr() + 4 * part_off, out.data_ptr() + 4 * out_offset, stride, S, K, layout, 0, T, Cin, out_stride, 0))

        red(0, dg, nbx, C, 2 * C, base=lnp)
        red(C, db, nbx, C, 2 * C, base=lnp)
        for y in range(ns):
            o = y * nbw * SL
            red(o, dw1, nbw, HS * C, SL, out_offset=y * HS * C)                                          # rows h0 .. h0+HS of [4C][C]
            red(o + HS * C, dw2, nbw, C * HS, SL, layout=2, T=Hd, Cin=HS, out_stride=1, out_offset=y * HS)   # columns h0 .. of [C][4C]
            red(o + 2 * HS * C, db1, nbw, HS, SL, out_offset=y * HS)
        red(2 * HS * C + HS, dbb2, nbw, C, SL)
        if all_sinks:
            _PENDING.extend(rows)
            return dx, None, None, None, None, None, None, None
        _reduce_now(rows, dev)
        return (dx, None if sg else dg, None if sb else db, None if s_w1 else dw1, None if s_b1 else db1, None if s_w2 else dw2,
                None if s_b2 else dbb2, None)


def wide_mlp_enabled(C, hidden, M=0):
    """Forward-only fused MLP half for the wide branches (C = 80 ... 320, weights streamed through LDS): pk_ln_mlp_wide_fwd.  With the
    token count M the library also says whether the launch would pay (enough workgroups).  POSE_FUSED_MLP_WIDE=0 switches it off
    (the unfused LayerNorm / fc1+GELU / fc2 sequence runs instead)."""
    return os.environ.get("POSE_FUSED_MLP_WIDE", "1") != "0" and bool(_lib.lib.pk_ln_mlp_wide_supported(C, hidden, 0 if _wide_forced() else M))


def _wide_forced():
    """POSE_FUSED_WIDE_FORCE=1: take the wide fused halves whenever they are built for the shape, however small the launch (tests)."""
    return os.environ.get("POSE_FUSED_WIDE_FORCE", "0") == "1"


def wide_attn_enabled(C, heads, n_windows=0):
    """Forward-only fused attention half of the head_dim-40 twins (pk_attn_block_wide_fwd); POSE_FUSED_ATTN_WIDE=0 switches it off."""
    return os.environ.get("POSE_FUSED_ATTN_WIDE", "1") != "0" and bool(
        _lib.lib.pk_attn_block_wide_supported(C, heads, 0 if _wide_forced() else n_windows))


def attn_half_wide_forward(x, g1, b1, table, wqkv, bqkv, wproj, bproj, scale1, heads, c_real=0, attn_scale=0.0):
    """x + s1 * proj(window_attention(qkv(LN1(x)))) in ONE launch for the 8-aligned twin of HRFormer-base (hrformer.py:262-286 at
    C = heads x 40): LayerNorm statistics over `c_real` channels, softmax scale `attn_scale` (the real head_dim^-0.5)."""
    wc = _wc()
    x = x.contiguous()
    B, H, W, C = x.shape
    amap, nwin = window_rowmap(B, H, W, x.device)
    y = _e((B, H, W, C), BF16, x.device)
    s1 = None if scale1 is None else scale1.float().contiguous()
    call("pk_attn_block_wide_fwd", x, amap, g1, b1, table, wc.fwd[id(wqkv)], bqkv, wc.fwd[id(wproj)], bproj, s1, y, B * nwin, nwin, heads, C,
         c_real or C, attn_scale or float(C // heads) ** -0.5, 1e-5, stream_ptr())
    return y


def mlp_half_wide_forward(x, g2, b2, w1, bias1, w2, bias2, scale2, c_real=0):
    """x + s2 * fc2(gelu(fc1(LN2(x)))) in ONE launch for inference at C = 80 / 128 / 160 / 256 / 320 (hrformer.py:287-293, Mlp :38-64)."""
    wc = _wc()
    x = x.contiguous()
    B, H, W, C = x.shape
    s2 = None if scale2 is None else scale2.float().contiguous()
    y = _e((B, H, W, C), BF16, x.device)
    call("pk_ln_mlp_wide_fwd", x, g2, b2, wc.fwd[id(w1)], bias1, wc.fwd[id(w2)], bias2, s2, y, B * H * W, C, c_real or C, w1.shape[0], H * W, 1e-5,
         stream_ptr())
    return y


def fused_attn_enabled(C, heads, c_real=0, attn_scale=0.0, train=True):
    """POSE_FUSED_ATTN (training) / POSE_FUSED_ATTN_EVAL (forward only) = channel counts that take the fused attention half."""
    ok = c_real in (0, C) and not attn_scale and bool(_lib.lib.pk_attn_block_supported(C, heads))
    # Measured in the captured training step (B = 64): the fused backward pays at C = 32 (3 076 img/s vs 3 003 with C = 64 fused as
    # well: its 2-head backward holds one wave per SIMD and 133 KB of LDS); forward-only use takes both widths.
    return ok and C in _fused_widths("POSE_FUSED_ATTN" if train else "POSE_FUSED_ATTN_EVAL", "32" if train else "32,64")


def attn_half_fused_forward(x, g1, b1, table, wqkv, bqkv, wproj, bproj, scale1, heads, save=False):
    """x + s1 * proj(window_attention(qkv(LN1(x)))) in ONE launch (pk_attn_block_fwd; C = 32 / 64, head_dim 32).
    -> (y, o, lse, rowmap); o / lse only when `save` (training)."""
    wc = _wc()
    x = x.contiguous()
    B, H, W, C = x.shape
    amap, nwin = window_rowmap(B, H, W, x.device)
    nw = B * nwin
    y = _e((B, H, W, C), BF16, x.device)
    o = _e((nw * WS * WS, C), BF16, x.device) if save else None
    lse = _e((nw * heads * WS * WS,), F32, x.device) if save else None
    s1 = None if scale1 is None else scale1.float().contiguous()
    call("pk_attn_block_fwd", x, amap, g1, b1, table, wc.fwd[id(wqkv)], bqkv, wc.fwd[id(wproj)], bproj, s1, y, o, lse, nw, nwin, heads, C,
         1e-5, stream_ptr())
    return y, o, lse, amap


class _AttnHalfFused(torch.autograd.Function):
    """x + s1 * proj(window_attention(qkv(LN1(x)))) as ONE forward launch (pk_attn_block_fwd) and one backward launch for the whole
    data-gradient chain (pk_attn_block_bwd: proj^T, attention core, qkv^T, LayerNorm backward, residual) that also emits dqkv and the
    window-ordered LayerNorm output for the two weight-gradient GEMMs.  Saved: x, the attention output o and the log-sum-exp."""

    @staticmethod
    def forward(ctx, x, g1, b1, table, wqkv, bqkv, wproj, bproj, scale1, heads):
        ctx.params = (g1, b1, table, wqkv, bqkv, wproj, bproj)
        wc = _wc()
        x = x.contiguous()
        y, o, lse, amap = attn_half_fused_forward(x, g1, b1, table, wqkv, bqkv, wproj, bproj, scale1, heads, save=True)
        s1 = None if scale1 is None else scale1.float().contiguous()
        ctx.save_for_backward(x, g1, b1, table, bqkv, s1 if s1 is not None else x.new_empty(0), o, lse, wc.fwd[id(wqkv)], wc.dgrad[id(wqkv)],
                              wc.dgrad[id(wproj)], amap)
        ctx.meta = (heads, s1 is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g1, b1, table, bqkv, s1, o, lse, wqkv_f, wqkv_t, wproj_t, amap = ctx.saved_tensors
        heads, has_s = ctx.meta
        s1 = s1 if has_s else None
        B, H, W, C = x.shape
        dev, M = dy.device, B * H * W
        nw = amap.numel() // (WS * WS)
        nwin, Mw = nw // B, nw * WS * WS
        dy = dy.contiguous()
        pg1, pb1, ptab, pwqkv, pbqkv, pwproj, pbproj = ctx.params
        (dg, sg), (db, sb) = _sink(pg1), _sink(pb1)
        # with_rpe=False (hrformer.py:145-191): the table is the shared all-zero constant, not a parameter -- its partial sums go to a
        # scratch workspace nobody reduces, so the LayerNorm rows keep the deferred (hipGraph-capturable) path
        no_tab = getattr(ptab, "_pk_zero_table", False)
        dtable, s_t = (None, True) if no_tab else _sink(ptab)
        nb = _lib.lib.pk_attn_block_blocks(nw)
        small_sinks = sg and sb and s_t and deferral_enabled()
        if small_sinks:
            lnp = _workspace(dg, "attn_ln", nb * 2 * C)
            rpb = _workspace(dg if no_tab else dtable, "attn_rpb_void" if no_tab else "attn_rpb", nb * 4 * heads * 169)
        else:
            lnp, rpb = _e((nb * 2 * C,), F32, dev), _e((nb * 4 * heads * 169,), F32, dev)
        dx = _e((B, H, W, C), BF16, dev)
        dqkv, u_w = _e((Mw, 3 * C), BF16, dev), _e((Mw, C), BF16, dev)
        call("pk_attn_block_bwd", dy, x, amap, g1, b1, table, wqkv_f, bqkv, wqkv_t, wproj_t, s1, o, lse, dx, dqkv, u_w, lnp, rpb, nw, nwin,
             heads, C, 1e-5, stream_ptr())
        rows = [(lnp.data_ptr(), dg.data_ptr(), 2 * C, nb, C, 0, 0, 1, 1, 1, 0), (lnp.data_ptr() + 4 * C, db.data_ptr(), 2 * C, nb, C, 0, 0, 1, 1, 1, 0)]
        for hh in range(0 if no_tab else heads):              # partial [wave][head][169] -> table[e][head]
            rows.append((rpb.data_ptr() + 4 * 169 * hh, dtable.data_ptr() + 4 * hh, 169 * heads, nb * 4, 169, 0, 0, 1, 1, heads, 0))
        if small_sinks:
            _PENDING.extend(rows)
        else:
            _reduce_now(rows, dev)
        # weight gradients: plain [tokens] x [features] GEMMs over the window-ordered rows
        dst, s_wq = _sink(pwqkv)
        dbqkv, s_bq = _sink(pbqkv)
        dwqkv = _wgrad(u_w, dqkv, 3 * C, C, 1, 1, None, M=Mw, out=dst, dbias=dbqkv, deferred=s_wq and s_bq)
        dst, s_wp = _sink(pwproj)
        dbproj, s_bp = _sink(pbproj)
        dwproj = _wgrad(o, dy.view(M, C), C, C, 1, 1, None, g_map=amap, g_scale=s1, g_rps=H * W, M=Mw, out=dst, dbias=dbproj,
                        deferred=s_wp and s_bp, win=(B, H, W))
        return (dx, None if sg else dg, None if sb else db, None if s_t else dtable, None if s_wq else dwqkv, None if s_bq else dbqkv,
                None if s_wp else dwproj, None if s_bp else dbproj, None, None)


def window_block(x, blk, heads, scale1=None, scale2=None):
    a = blk.attn
    c_real, attn_scale = getattr(blk, "c_real", 0), getattr(blk, "attn_scale", 0.0)      # set on padded twins (models/padded.py)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or blk.norm1.weight.requires_grad)
    if not needs_grad and wide_attn_enabled(x.shape[-1], heads, x.shape[0] * -(-x.shape[1] // WS) * -(-x.shape[2] // WS)):
        x = attn_half_wide_forward(x, blk.norm1.weight, blk.norm1.bias, rel_table(a, heads), a.qkv.weight, a.qkv.bias, a.proj.weight,
                                   a.proj.bias, scale1, heads, c_real, attn_scale)
    elif fused_attn_enabled(x.shape[-1], heads, c_real, attn_scale, train=needs_grad):
        if needs_grad:
            x = _AttnHalfFused.apply(x, blk.norm1.weight, blk.norm1.bias, rel_table(a, heads), a.qkv.weight, a.qkv.bias,
                                     a.proj.weight, a.proj.bias, scale1, heads)
        else:
            x = attn_half_fused_forward(x, blk.norm1.weight, blk.norm1.bias, rel_table(a, heads), a.qkv.weight, a.qkv.bias,
                                        a.proj.weight, a.proj.bias, scale1, heads)[0]
    else:
        x = _AttnHalf.apply(x, blk.norm1.weight, blk.norm1.bias, rel_table(a, heads), a.qkv.weight, a.qkv.bias,
                            a.proj.weight, a.proj.bias, scale1, heads, c_real, attn_scale)
    m = blk.mlp
    if not needs_grad and wide_mlp_enabled(x.shape[-1], m.fc1.weight.shape[0], x.numel() // x.shape[-1]):
        return mlp_half_wide_forward(x, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, scale2, c_real)
    if fused_mlp_enabled(x.shape[-1], c_real) and m.fc1.weight.shape[0] == 4 * x.shape[-1]:
        return _MlpHalfFused.apply(x, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, scale2)
    return _MlpHalf.apply(x, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, scale2, c_real)


# ================================================================================================ exchange unit
class _FuseSum(torch.autograd.Function):
    """relu(sum_i up(x_i)) with bilinear (align_corners=False) up-sampling of the lower-resolution inputs fused in."""

    @staticmethod
    def forward(ctx, relu, *xs):
        xs = [t.contiguous() for t in xs]
        ref = max(xs, key=lambda t: t.shape[1] * t.shape[2])
        B, H, W, C = ref.shape
        dev = ref.device
        ptrs = torch.tensor([t.data_ptr() for t in xs], dtype=torch.int64)
        hs = torch.tensor([t.shape[1] for t in xs], dtype=torch.int32)
        wsz = torch.tensor([t.shape[2] for t in xs], dtype=torch.int32)
        out = _e((B, H, W, C), BF16, dev)
        # host arrays: the C-ABI reads them before the launch returns (pointer table is copied into the kernel arguments)
        call("pk_fuse_sum", ptrs.data_ptr(), hs.data_ptr(), wsz.data_ptr(), len(xs), out, B, H, W, C, 1 if relu else 0, stream_ptr())
        ctx.save_for_backward(out)
        ctx.meta = (relu, [tuple(t.shape) for t in xs])
        return out

    @staticmethod
    def backward(ctx, dy):
        (out,) = ctx.saved_tensors
        relu, shapes = ctx.meta
        B, H, W, C = out.shape
        dy = dy.contiguous()
        g = dy
        if relu:
            g = _e(out.shape, BF16, out.device)
            call("pk_relu_bwd", dy, out, g, out.numel(), stream_ptr())
        grads = []
        for shp in shapes:
            if shp[1] == H and shp[2] == W:
                grads.append(g)
            else:
                d = _e(shp, BF16, out.device)
                call("pk_upsample_bilinear_bwd", g, d, B, H, W, shp[1], shp[2], C, stream_ptr())
                grads.append(d)
        return (None, *grads)


def sum_same_shape(ts):
    """Plain sum of 2-4 NHWC bf16 tensors of one shape in ONE launch (no autograd): used for fan-out gradient accumulation."""
    ts = [t.contiguous() for t in ts]
    B, H, W, C = ts[0].shape
    out = _e((B, H, W, C), BF16, ts[0].device)
    ptrs = torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64)
    hs = torch.tensor([H] * len(ts), dtype=torch.int32)
    wsz = torch.tensor([W] * len(ts), dtype=torch.int32)
    call("pk_fuse_sum", ptrs.data_ptr(), hs.data_ptr(), wsz.data_ptr(), len(ts), out, B, H, W, C, 0, stream_ptr())
    return out


def fuse_sum(xs, relu=True):
    return _FuseSum.apply(relu, *xs)


def exchange(xs, fuse, training, n_out=None, first_only=False):
    """Exchange unit (hrformer.py:462-491 == hrnet.py:198-227): out_i = relu(sum_j route_{j->i}(x_j)).

    One parallel task per OUTPUT.  (Measured and dropped: one task per ROUTE j -> i, i.e. 12 shorter chains plus a second region for
    the sums -- 19.15 -> 19.5 ms per step: hipGraph runs parallel branches on four hardware queues, more branches only add fork /
    join edges; and packing the routes into four tasks of equal layer count (4 + 4 + 4 + 4 instead of 3 + 3 + 4 + 6) with the sums
    afterwards -- 18.9 -> 19.2 .. 19.5 ms: every x_j then also gets a gradient from outside the region, i.e. four extra elementwise
    adds per unit on the main stream.  The unit stays a chain of launch-latency-bound kernels on small tensors: ~0.3 ms backward per unit with ~0.3
    kernels in flight, profiles/r02_trace_summary.txt.)"""
    from . import dispatch, exchange as xg
    n = len(xs)
    if xg.usable(training, xs) and 1 < n <= 4 and (first_only or xg.whole_unit()):
        # grouped launches: one launch per kernel family and dependency level of the unit (exchange.py)
        if first_only:
            y0 = xg.unit(xs, fuse, training, outs=[0])
            if training:        # the unused outputs (see below): BatchNorm running statistics only, off the critical path
                det = [t.detach() for t in xs]
                dispatch.run_detached(lambda: xg.unit(det, fuse, training, outs=range(1, n)), det)
            return y0
        return xg.unit(xs, fuse, training, outs=range(n if n_out is None else n_out))
    if first_only and n > 1:
        # Last module of the network: only output 0 is consumed (hrformer.py:776 / hrnet.py:441).  The reference still computes outputs
        # 1..n-1; their only lasting effect is the running-statistics update of their BatchNorm layers in training mode.  They leave the
        # critical path: eval mode skips them, training runs them without autograd on a detached side stream that is joined when the
        # outermost dispatch.scope exits (they used to be three branches of a fork / join in front of the head: 164 us vs 70 us for output 0).
        y0 = exchange_output(0, xs, fuse, training)
        if training:
            det = [t.detach() for t in xs]
            dispatch.run_detached(lambda: [exchange_output(i, det, fuse, training) for i in range(1, n)], det)
        return [y0]
    n_o = n if n_out is None else n_out
    return dispatch.parallel([(lambda ins, i=i: exchange_output(i, ins, fuse, training)) for i in range(n_o)], [list(xs)] * n_o)


def exchange_output(i, xs, fuse, training):
    """Output i of an exchange unit: relu(sum_j route_{j->i}(x_j)).  Also called from inside the NEXT module's branch task i
    (models: chained modules), so that branch i starts as soon as ITS input is ready instead of after the slowest output."""
    from . import exchange as xg
    if xg.usable(training, xs) and 1 < len(xs) <= 4:
        # the routes into output i as grouped launches: one launch per kernel family and chain level (exchange.py)
        return xg.unit(list(xs), fuse, training, outs=[i])[0]
    terms = []
    for j in range(len(xs)):
        if j == i:
            terms.append(xs[j])
        elif j > i:
            conv, bn = fuse[str(i)][str(j)]
            terms.append(conv_bn_act(xs[j], conv, bn, False, None, training))      # up-sampled inside fuse_sum
        else:
            t = xs[j]
            chain = fuse[str(i)][str(j)]
            for s, (conv, bn) in enumerate(chain):
                t = conv_bn_act(t, conv, bn, s != len(chain) - 1, None, training)
            terms.append(t)
    return fuse_sum(terms, True)


def drop_scales(n_draws, batch, drop_prob, device):
    """All DropPath multipliers of one step in one launch: floor(keep + U)/keep, shape (n_draws, B) (hrformer.py:15-24)."""
    keep = 1.0 - drop_prob
    return torch.floor(keep + torch.rand(n_draws, batch, device=device)) / keep


# ================================================================================================ backend choice
def supported(model) -> bool:
    """True when every conv / linear of `model` fits the HIP kernels (channels % 8 == 0, head_dim a multiple of 8 up to 64)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d):
            if m.weight.shape[0] % 8 and m.weight.shape[2] != 1:
                return False
            if m.weight.shape[1] % 8 and m.weight.shape[1] != 3:
                return False
        if hasattr(m, "qkv") and hasattr(m, "num_heads"):
            d = m.qkv.weight.shape[1] // m.num_heads
            if d > 64 or d % 8:
                return False
        if isinstance(m, torch.nn.LayerNorm) and m.normalized_shape[0] % 8:
            return False
    return True
