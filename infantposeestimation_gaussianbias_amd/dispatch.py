"""Per-model backend selection for the network ops.

`with dispatch.scope(model):` activates, for the duration of a forward pass, either `nnops` (hand-written HIP
kernels, NHWC bf16 tensors, bf16 weight cache refreshed on entry) when every layer of `model` fits the kernels, or
`nnops_aten` (PyTorch-ROCm composites, channels_last tensors) otherwise.  Nested scopes reuse the outer one, so a
backbone called from PoseEstimator shares its weight cache.  `backend_name(model)` reports the choice.
"""
import contextlib

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
