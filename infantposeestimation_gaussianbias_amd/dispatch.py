"""Entry to the network ops (`nnops`: hand-written HIP kernels, NHWC bf16 tensors).

`with dispatch.scope(model):` activates the kernels for the duration of a forward pass and refreshes the model's bf16 weight
cache on entry; nested scopes reuse the outer one, so a backbone called from PoseEstimator shares its cache.  A model whose
channel counts are not multiples of 8 (HRFormer-base, HRNet-W18) is run through its 8-aligned padded twin
(`padded_twin`, models/padded.py); a model that fits neither raises -- there is no PyTorch or CPU fallback.
"""
import contextlib
import os

import torch

from . import nnops
from ._lib import PoseKernelError

_ACTIVE = []


def backend_for(model):
    ok = getattr(model, "_pk_supported", None)
    if ok is None:
        ok = nnops.supported(model)
        object.__setattr__(model, "_pk_supported", ok)
    if not ok:
        raise PoseKernelError(
            f"{type(model).__name__}: layer shapes outside the HIP kernels (channels must be multiples of 8, head_dim a multiple of 8 "
            "up to 64, window 7) and no 8-aligned padded twin could be built (models/padded.py); there is no fallback path")
    return nnops


def padded_twin(model):
    """8-aligned twin of a model whose channel counts the HIP kernels cannot take (models/padded.py), else None."""
    if getattr(model, "_pk_twin", None) is False:
        return None
    from .models import padded
    return padded.twin_for(model)


def backend_name(model) -> str:
    if torch.cuda.is_available() and padded_twin(model) is not None:
        return "hip (8-aligned padded twin)"
    backend_for(model)
    return "hip"


@contextlib.contextmanager
def scope(model):
    if _ACTIVE:
        yield _ACTIVE[-1]
        return
    ops = backend_for(model)
    _ACTIVE.append(ops)
    try:
        with nnops.use_weights(model):
            yield ops
            join_detached()
    finally:
        _ACTIVE.pop()


def ops():
    if not _ACTIVE:
        raise RuntimeError("network op called outside dispatch.scope(model)")
    return _ACTIVE[-1]


# the network ops themselves (they refuse to run outside a scope: nnops._wc() raises without an active weight cache)
from .nnops import (backward_milestone, conv_bn_act, deconv_bn_relu, drop_scales, exchange, exchange_output, head_out, mlp_rows, residual_block,  # noqa: E402,F401
                    to_features, window_attention_tokens, window_block)


def to_public(x):
    """Internal feature map -> the reference's (B,C,H,W) view (no copy)."""
    ops()
    return x.permute(0, 3, 1, 2)


def from_public(x):
    """(B,C,H,W) feature map produced by one of our backbones (bf16) or a raw fp32 tensor -> internal layout."""
    ops()
    if x.dtype != nnops.ACT_DTYPE:
        B, C, H, W = x.shape
        if x.requires_grad and torch.is_grad_enabled() and C % 8 == 0:
            # a leaf module called on its own with a differentiable fp32 tensor (public surface): keep autograd's chain through the cast
            return x.permute(0, 2, 3, 1).to(nnops.ACT_DTYPE).contiguous()
        return nnops.to_features(x, cpad=-(-C // 8) * 8)
    return x.permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------------------------------------- branch concurrency
# The 2-4 resolution branches of an HRNet/HRFormer module are independent between exchange units, and only branch 0
# (64x48 at B=64) has enough workgroups to fill 256 CUs; the others are latency-bound launches of < 256 workgroups.
# `parallel` runs callable 0 on the current stream and the others on per-device side streams (fork: side.wait(current);
# join: current.wait(side)), so the small branches hide under the big one.  Two implementations:
#   * eager (default): the callables run under `torch.cuda.stream(side)`; autograd replays each node on the stream its
#     forward ran on and inserts its own cross-stream events in backward;
#   * region mode (`set_region_mode(True)`, used for hipGraph capture): the whole fork/join is ONE autograd node.  Its
#     forward builds a private autograd graph per callable on that callable's stream; its backward forks again, runs
#     each private graph with a nested backward on its own stream and joins.  From the outer engine's point of view
#     everything happens on one stream, so the captured graph has the same star-shaped fork/join in forward and
#     backward.  (Capturing the eager variant crashes hipStreamEndCapture on ROCm 7.2 - scripts/gpu_graph_streams.py:
#     the engine's direct side-stream <-> side-stream event edges in backward are the difference.)
# POSE_STREAMS=0 disables both.
_SIDE = {}
_STREAMS_OFF = [False]
_REGION = [False]
_EVENTS = []               # events used for fork/join stay alive: a captured graph may refer to them until the capture ends


def set_streams(enabled: bool):
    _STREAMS_OFF[0] = not enabled


def set_region_mode(enabled: bool):
    _REGION[0] = bool(enabled)


def streams_enabled() -> bool:
    return (not _STREAMS_OFF[0]) and os.environ.get("POSE_STREAMS", "1") != "0" and torch.cuda.is_available()


def _side_stream(dev, i):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), i)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def join_side_streams(cur=None):
    """Make `cur` wait for everything enqueued so far on every side stream."""
    cur = cur or torch.cuda.current_stream()
    for s in _SIDE.values():
        cur.wait_stream(s)


_DETACHED = []


def run_detached(fn, tensors):
    """Run fn() without autograd on a dedicated side stream that nothing waits for until the outermost `scope` exits (`join_detached`):
    work whose results nobody reads in this pass (the unused exchange outputs of the last module: only their BatchNorm running-statistics
    updates survive).  `tensors`: what fn reads."""
    with torch.no_grad():
        if not streams_enabled() or _MARK:
            fn()
            return
        cur = torch.cuda.current_stream()
        s = _side_stream(cur.device, 7)
        _order(cur, s)
        with torch.cuda.stream(s):
            for t in tensors:
                t.record_stream(s)
            fn()
        _DETACHED.append(s)


def join_detached():
    if _DETACHED:
        cur = torch.cuda.current_stream()
        for s in _DETACHED:
            _order(s, cur)
        _DETACHED.clear()


# (Measured and removed: issuing the weight-gradient kernels on auxiliary streams -- they are off the data-gradient dependency
# chain -- made the captured step SLOWER, 25.4 -> 31.0 ms for all of them and -1 % at best for the head's five large ones: their
# >= 2 048-workgroup grids take CUs from the critical chain instead of filling its bubbles, and hipGraph maps parallel branches
# onto 4 hardware queues (DEBUG_HIP_FORCE_GRAPH_QUEUES; more queues do not help).  Round 3, again for the SERIAL phases only (stem, layer1,
# transitions: one kernel in flight, the weight gradients of >= 100 000-row convs on one auxiliary stream joined by finalize_deferred):
# 17.16 vs 17.21 ms over four alternating runs each -- inside the noise, not kept.  Round 4, only the slab launches of the LOW-resolution
# branches (i >= 2: latency-bound chains, the longest branches of HRNet-W32's regions when run alone), one shared or one auxiliary
# stream per branch, operands held until the region joins: cfg 4 20.28 -> 24.17 / 23.96 ms, cfg 2 16.15 -> 18.3 ms
# (scripts/gpu_r04_n.sh) -- every launch moved aside costs two cross-queue event edges in the captured graph, more than it hides.)


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)


def _order(src, dst):
    """dst waits for everything enqueued on src so far."""
    ev = torch.cuda.Event()
    ev.record(src)
    dst.wait_event(ev)
    if torch.cuda.is_current_stream_capturing():
        _EVENTS.append(ev)
    elif _EVENTS:
        _EVENTS.clear()


_MARK = os.environ.get("POSE_MARKERS", "0") == "1"           # profiling: section markers + branches serialised on one stream


def _mark(i):
    if _MARK:
        from ._lib import call, stream_ptr
        call("pk_marker", i, stream_ptr())


class _Region(torch.autograd.Function):
    """fork -> fns[i](inputs_i) on stream i -> join, as one autograd node (see the section comment)."""

    @staticmethod
    def forward(ctx, fns, counts, *flat):
        cur = torch.cuda.current_stream()
        n = len(fns)
        side = [cur] + [cur if _MARK else _side_stream(cur.device, i) for i in range(1, n)]
        for i in range(1, n):
            _order(cur, side[i])
        graphs, pos = [], 0
        for i in range(n):
            ins = flat[pos:pos + counts[i]]
            pos += counts[i]
            with torch.cuda.stream(side[i]), torch.enable_grad():
                _mark(100 + i)
                loc = []
                for t in ins:
                    if i:
                        t.record_stream(side[i])
                    loc.append(t.detach().requires_grad_(True) if (t.requires_grad and t.is_floating_point()) else t)
                out = fns[i](loc)
            if not torch.is_tensor(out):
                raise TypeError("parallel(): every callable must return one tensor")
            graphs.append((loc, [out]))
        for i in range(1, n):
            _order(side[i], cur)
            for t in graphs[i][1]:
                t.record_stream(cur)
        _mark(99)
        ctx.graphs, ctx.side = graphs, side
        first = {}
        ctx.same = [first.setdefault(id(t), k) for k, t in enumerate(flat)]     # index of the first occurrence of each input
        return tuple(t.detach() for _, outs in graphs for t in outs)

    @staticmethod
    def backward(ctx, *gouts):
        cur = torch.cuda.current_stream()
        side = [cur] + list(ctx.side[1:])
        n = len(side)
        for i in range(1, n):
            _order(cur, side[i])
        grads, pos = [], 0
        for i, (loc, outs) in enumerate(ctx.graphs):
            gs = gouts[pos:pos + len(outs)]
            pos += len(outs)
            pairs = [(o, g) for o, g in zip(outs, gs) if g is not None and o.requires_grad]
            with torch.cuda.stream(side[i]):
                _mark(200 + i)
                if pairs:
                    if i:
                        for _, g in pairs:
                            g.record_stream(side[i])
                    torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
            for t in loc:
                g = t.grad if (t.requires_grad and t.is_leaf) else None
                if g is not None and i:
                    g.record_stream(cur)
                grads.append(g)
                if g is not None:
                    t.grad = None
        for i in range(1, n):
            _order(side[i], cur)
        _mark(199)
        ctx.graphs = None
        # An input shared by several callables (exchange unit: every output reads every branch) gets one gradient per use.
        # Sum them here in ONE launch per input instead of leaving 2-3 elementwise adds per input to the autograd engine.
        groups = {}
        for k, f in enumerate(ctx.same):
            if grads[k] is not None:
                groups.setdefault(f, []).append(k)
        for f, ks in groups.items():
            if len(ks) < 2:
                continue
            gs = [grads[k] for k in ks]
            if len(gs) <= 4 and gs[0].dim() == 4 and gs[0].dtype == nnops.ACT_DTYPE and all(g.shape == gs[0].shape for g in gs):
                total = nnops.sum_same_shape(gs)
                for k in ks:
                    grads[k] = None
                grads[ks[0]] = total
        return (None, None, *grads)


def parallel(fns, inputs):
    """Run fns[0](inputs[0]) on the current stream and fns[i>0](inputs[i]) concurrently on side streams; returns the
    results (one tensor per callable) in order.  `inputs[i]` is the list of tensors fns[i] reads (everything else it
    touches must be parameters or tensors it creates itself)."""
    n = len(fns)
    if n == 1 or not streams_enabled():
        return [f(list(x)) for f, x in zip(fns, inputs)]
    if _REGION[0] and torch.is_grad_enabled():
        counts = [len(x) for x in inputs]
        flat = [t for x in inputs for t in x]
        return list(_Region.apply(fns, counts, *flat))
    if torch.is_grad_enabled() and torch.cuda.is_current_stream_capturing():
        # Capturing the eager multi-stream schedule makes autograd insert direct side-stream <-> side-stream event edges in
        # backward, and hipStreamEndCapture segfaults on them (ROCm 7.2; bisected in scripts/gpu_graph_streams.py).
        raise PoseKernelError("hipGraph capture with concurrent branch streams needs region mode: call dispatch.set_region_mode(True) "
                              "(engine.Trainer(graph_streams=True) does) or dispatch.set_streams(False) before capturing")
    cur = torch.cuda.current_stream()
    outs = [None] * n
    side = [None] + [_side_stream(cur.device, i) for i in range(1, n)]
    for i in range(1, n):                     # fork first: side streams depend only on work enqueued before this point
        side[i].wait_stream(cur)
        for t in inputs[i]:
            t.record_stream(side[i])
    outs[0] = fns[0](list(inputs[0]))         # then run in index order, so autograd nodes are created exactly as in the
    for i in range(1, n):                     # sequential schedule (same backward order, same bf16 accumulation order)
        with torch.cuda.stream(side[i]):
            outs[i] = fns[i](list(inputs[i]))
    for i in range(1, n):
        cur.wait_stream(side[i])
        for t in _tensors(outs[i]):
            t.record_stream(cur)
    return outs
