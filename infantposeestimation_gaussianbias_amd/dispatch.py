"""Per-model backend selection for the network ops.

`with dispatch.scope(model):` activates, for the duration of a forward pass, either `nnops` (hand-written HIP
kernels, NHWC bf16 tensors, bf16 weight cache refreshed on entry) when every layer of `model` fits the kernels, or
`nnops_aten` (PyTorch-ROCm composites, channels_last tensors) otherwise.  Nested scopes reuse the outer one, so a
backbone called from PoseEstimator shares its weight cache.  `backend_name(model)` reports the choice.
"""
import contextlib
import os

import torch

from . import nnops, nnops_aten

_ACTIVE = []


def backend_for(model):
    name = getattr(model, "_pk_backend", None)
    if name is None:
        name = "hip" if nnops.supported(model) else "aten"
        object.__setattr__(model, "_pk_backend", name)
    return nnops if name == "hip" else nnops_aten


def backend_name(model) -> str:
    backend_for(model)
    return model._pk_backend


@contextlib.contextmanager
def scope(model):
    if _ACTIVE:
        yield _ACTIVE[-1]
        return
    ops = backend_for(model)
    _ACTIVE.append(ops)
    try:
        if ops is nnops:
            with nnops.use_weights(model):
                yield ops
        else:
            yield ops
    finally:
        _ACTIVE.pop()


def ops():
    if not _ACTIVE:
        raise RuntimeError("network op called outside dispatch.scope(model)")
    return _ACTIVE[-1]


def to_features(x):
    return ops().to_features(x)


def conv_bn_act(x, conv, bn, relu=False, residual=None, training=False):
    return ops().conv_bn_act(x, conv, bn, relu, residual, training)


def head_out(x, conv, softplus=False):
    return ops().head_out(x, conv, softplus)


def window_block(x, blk, heads, scale1=None, scale2=None):
    return ops().window_block(x, blk, heads, scale1, scale2)


def exchange(xs, fuse, training, n_out=None):
    return ops().exchange(xs, fuse, training, n_out)


def drop_scales(n_draws, batch, drop_prob, device):
    return nnops.drop_scales(n_draws, batch, drop_prob, device)


def to_public(x):
    """Internal feature map -> the reference's (B,C,H,W) view (no copy)."""
    return x.permute(0, 3, 1, 2) if ops() is nnops else x


def from_public(x):
    """(B,C,H,W) feature map produced by one of our backbones (bf16) or a raw fp32 tensor -> internal layout."""
    if x.dtype != nnops.ACT_DTYPE:
        if ops() is nnops:
            B, C, H, W = x.shape
            return nnops.to_features(x, cpad=-(-C // 8) * 8)
        return nnops_aten.to_features(x)
    return x.permute(0, 2, 3, 1).contiguous() if ops() is nnops else x


# ---------------------------------------------------------------------------------------------- branch concurrency
# The 2-4 resolution branches of an HRNet/HRFormer module are independent between exchange units, and only branch 0
# (64x48 at B=64) has enough workgroups to fill 256 CUs; the others are latency-bound launches of < 256 workgroups.
# `parallel` runs callable 0 on the current stream and the others on per-device side streams (fork: side.wait(current);
# join: current.wait(side)), so the small branches hide under the big one.  Backward follows automatically: autograd
# replays each node on the stream its forward ran on and joins at the end.  Under hipGraph capture the fork/join
# becomes parallel graph branches.  POSE_STREAMS=0 disables it.
_SIDE = {}
_STREAMS_OFF = [False]     # set by engine.Trainer(use_graph=True): multi-stream capture crashes hipStreamEndCapture on ROCm 7.2


def set_streams(enabled: bool):
    _STREAMS_OFF[0] = not enabled


def streams_enabled() -> bool:
    return (not _STREAMS_OFF[0]) and os.environ.get("POSE_STREAMS", "1") != "0" and torch.cuda.is_available()


def _side_stream(dev, i):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), i)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)


def parallel(fns, inputs=None):
    """Run fns[0] on the current stream and fns[1:] concurrently on side streams; returns their results in order.
    `inputs[i]` (tensor / list of tensors) are the tensors fns[i] reads that were produced on other streams."""
    n = len(fns)
    if n == 1 or not streams_enabled() or ops() is not nnops:
        return [f() for f in fns]
    cur = torch.cuda.current_stream()
    outs = [None] * n
    side = [None] + [_side_stream(cur.device, i) for i in range(1, n)]
    for i in range(1, n):                     # fork first: side streams depend only on work enqueued before this point
        side[i].wait_stream(cur)
        if inputs is not None:
            for t in _tensors(inputs[i]):
                t.record_stream(side[i])
    outs[0] = fns[0]()                        # then run in index order, so autograd nodes are created exactly as in the
    for i in range(1, n):                     # sequential schedule (same backward order, same bf16 accumulation order)
        with torch.cuda.stream(side[i]):
            outs[i] = fns[i]()
    for i in range(1, n):
        cur.wait_stream(side[i])
        for t in _tensors(outs[i]):
            t.record_stream(cur)
    return outs
