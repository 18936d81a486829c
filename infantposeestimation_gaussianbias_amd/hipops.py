"""Host-side wrappers over the C-ABI: argument checking, output allocation (torch owns device memory),
and `torch.autograd.Function`s for the differentiable ops.  Every function here launches HIP kernels from
libposekernels.so on the current torch stream; none has a PyTorch fallback.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import call, stream_ptr

F32, I32 = torch.float32, torch.int32


def _chk(t, dtype=F32, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.PoseKernelError(f"{name}: expected a CUDA(HIP) tensor; the hot path has no CPU implementation")
    if t.dtype != dtype:
        raise _lib.PoseKernelError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


# ------------------------------------------------------------------------------------------------ T1 / T2
_LUT_CACHE = {}


def _call_if(n, name, *args):
    """Launch unless the batch is empty: the reference's tensor code returns empty results for B = 0 (a frame without detections), and a
    zero-size grid is a launch error.  Nothing is computed on the host either way: the outputs are already the right (empty) shape."""
    if n > 0:
        call(name, *args)


def _patch_lut(sigma: float, device):
    """Host-built LUT of the reference's float32 patch (datasets/coco_dataset.py:227-233), indexed by d²."""
    key = (float(sigma), str(device))
    if key not in _LUT_CACHE:
        size = 2 * (sigma * 3) + 1
        ax = np.arange(0, size, 1, np.float32)
        c = size // 2
        g = np.exp(-((ax[None, :] - c) ** 2 + (ax[:, None] - c) ** 2) / (2 * sigma ** 2))   # same float32 expression
        n, ci = g.shape[0], int(c)
        far = max(ci, n - 1 - ci)
        lut = np.zeros(2 * far * far + 1, np.float32)
        for j in range(n):
            for i in range(n):
                lut[(i - ci) ** 2 + (j - ci) ** 2] = g[j, i]
        _LUT_CACHE[key] = (torch.from_numpy(lut).to(device), n, ci)
    return _LUT_CACHE[key]


def gaussian_target(keypoints, visible, input_size, heatmap_size, sigma):
    """(B,K,2),(B,K) -> target (B,K,Hh,Wh), weight (B,K,1). Sizes are (W,H) as in the reference config."""
    kp, vis = _chk(keypoints, name="keypoints"), _chk(visible, name="visible")
    B, K = vis.shape
    wh, hh = int(heatmap_size[0]), int(heatmap_size[1])
    lut, n, c = _patch_lut(float(sigma), kp.device)
    target = torch.empty(B, K, hh, wh, dtype=F32, device=kp.device)
    weight = torch.empty(B, K, 1, dtype=F32, device=kp.device)
    _call_if(B * K, "pk_gaussian_target", kp, vis, lut, lut.numel(), target, weight, B, K, hh, wh,
         float(input_size[0]) / float(heatmap_size[0]), float(input_size[1]) / float(heatmap_size[1]), float(sigma) * 3, n, c,
         stream_ptr())
    return target, weight


def dense_target(keypoints, visible, input_size_hw, heatmap_size_hw, sigma):
    kp, vis = _chk(keypoints, name="keypoints"), _chk(visible, name="visible")
    B, K = vis.shape
    hh, hw = int(heatmap_size_hw[0]), int(heatmap_size_hw[1])
    hm = torch.empty(B, K, hh, hw, dtype=F32, device=kp.device)
    w = torch.empty(B, K, dtype=F32, device=kp.device)
    _call_if(B * K, "pk_dense_target", kp, vis, hm, w, B, K, hh, hw, float(np.float32(hw / input_size_hw[1])),
         float(np.float32(hh / input_size_hw[0])), float(sigma), stream_ptr())
    return hm, w


# ------------------------------------------------------------------------------------------------ decoders
def argmax_decode(heatmaps, mode=0):
    """-> index (B,K) int32, maxval (B,K), coords (B,K,2). mode 0 plain / 1 quarter-shift / 2 Taylor."""
    hm = _chk(heatmaps, name="heatmaps")
    B, K, H, W = hm.shape
    idx = torch.empty(B, K, dtype=I32, device=hm.device)
    mv = torch.empty(B, K, dtype=F32, device=hm.device)
    co = torch.empty(B, K, 2, dtype=F32, device=hm.device)
    _call_if(B * K, "pk_argmax_decode", hm, idx, mv, co, B * K, H, W, int(mode), stream_ptr())
    return idx, mv, co


def softargmax_refine_decode(heatmaps, offsets, alpha_param, fusion_weight_param, radius=2):
    hm = _chk(heatmaps, name="heatmaps")
    B, K, H, W = hm.shape
    off = None if offsets is None else _chk(offsets, name="offsets")
    co = torch.empty(B, K, 2, dtype=F32, device=hm.device)
    sc = torch.empty(B, K, dtype=F32, device=hm.device)
    _call_if(B * K, "pk_softargmax_refine_decode", hm, off, _chk(alpha_param.reshape(1)), None if off is None else _chk(fusion_weight_param.reshape(1)),
         co, sc, B * K, H, W, int(radius), stream_ptr())
    return co, sc


def window_refine(heatmaps, coords, window=5):
    hm, c = _chk(heatmaps), _chk(coords)
    B, K, H, W = hm.shape
    out = torch.empty_like(c)
    _call_if(B * K, "pk_window_refine", hm, c, out, B * K, H, W, int(window), stream_ptr())
    return out


def fused_blend(hp, maxvals, regression, sx, sy, reg_scale):
    hp = _chk(hp)
    out = torch.empty_like(hp)
    _call_if(hp.shape[0] * hp.shape[1], "pk_fused_blend", hp, None if maxvals is None else _chk(maxvals), None if regression is None else _chk(regression), out,
         hp.shape[0] * hp.shape[1], float(sx), float(sy), float(reg_scale), stream_ptr())
    return out


def affine_coords(coords, center, scale, mul_x, mul_y, mask_maxvals=None, threshold=0.0):
    c = _chk(coords)
    B, K = c.shape[:2]
    out = torch.empty_like(c)
    _call_if(B * K, "pk_affine_coords", c, _chk(center), _chk(scale), out, B, K, float(mul_x), float(mul_y),
         None if mask_maxvals is None else _chk(mask_maxvals), float(threshold), stream_ptr())
    return out


def pose_records(keypoints, scores):
    """(B,K,2), (B,K) -> records (B,K,3) = [x, y, score], instance score (B,) = mean of the positive scores (COCOEvaluator.update)."""
    kp, sc = _chk(keypoints, name="keypoints"), _chk(scores, name="scores")
    B, K = sc.shape
    rec = torch.empty(B, K, 3, dtype=F32, device=kp.device)
    inst = torch.empty(B, dtype=F32, device=kp.device)
    _call_if(B * K, "pk_pose_records", kp, sc, rec, inst, B, K, stream_ptr())
    return rec, inst


def flip_merge(hm, hm_from_flipped, partner):
    a, b = _chk(hm), _chk(hm_from_flipped)
    B, K, H, W = a.shape
    out = torch.empty_like(a)
    _call_if(B * K, "pk_flip_merge", a, b, _chk(partner, I32), out, B, K, H, W, stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------ losses
def loss_ws_floats(B, K):
    return B * K * 24 + B * 16 * 4 + 16


class _FusionLoss(torch.autograd.Function):
    """L1/L2 forward + hand-derived backward (fusion_head.py:745-806). Returns the 7 loss values as one tensor."""

    @staticmethod
    def forward(ctx, hm, off, var, target, weight, gt, in_w, in_h, sigma_t, lambdas, use_target_weight=True):
        hm, off, var = _chk(hm, name="heatmaps"), _chk(off, name="offsets"), _chk(var, name="variances")
        target, weight, gt = _chk(target), _chk(weight), _chk(gt)
        B, K, H, W = hm.shape
        if off.shape != (B, K, 2, H, W) or var.shape != hm.shape or target.shape != hm.shape or weight.numel() != B * K or gt.shape != (B, K, 2):
            raise _lib.PoseKernelError("fusion loss: inconsistent shapes")
        if lambdas.numel() == 6:           # term weights only: default overlap threshold (fusion_head.py:404)
            lambdas = torch.cat([lambdas.float(), lambdas.new_full((1,), 0.5, dtype=F32)])
        lambdas = _chk(lambdas.float(), name="lambdas")
        ws = torch.empty(loss_ws_floats(B, K), dtype=F32, device=hm.device)
        losses = torch.empty(7, dtype=F32, device=hm.device)
        call("pk_fusion_loss_fwd", hm, off, var, target, weight, gt, ws, losses, B, K, H, W, float(in_w), float(in_h), float(sigma_t),
             lambdas, 1 if use_target_weight else 0, stream_ptr())
        ctx.save_for_backward(hm, off, var, target, weight, ws, lambdas)
        ctx.sigma_t = float(sigma_t)
        return losses

    @staticmethod
    def backward(ctx, g):
        hm, off, var, target, weight, ws, lambdas = ctx.saved_tensors
        B, K, H, W = hm.shape
        # d(sum_i g_i * loss_i): entries 0..5 are the weighted terms and entry 6 their sum, so the per-term
        # multipliers are lambdas*(g[:6]+g[6]); the kernel takes them as "lambdas" with grad_total = 1.
        # (the kernel applies them: five tiny ATen launches at the very start of backward otherwise)
        dhm, doff, dvar = torch.empty_like(hm), torch.empty_like(off), torch.empty_like(var)
        call("pk_fusion_terms_bwd", hm, target, ws, None, None, _chk(g.float()), dhm, doff, dvar, None, B, K, H, W, ctx.sigma_t, lambdas,
             stream_ptr())
        return dhm, doff, dvar, None, None, None, None, None, None, None, None


def fusion_loss(hm, off, var, target, weight, gt, input_size, sigma_t, lambdas_dev, use_target_weight=True):
    return _FusionLoss.apply(hm, off, var, target, weight, gt, float(input_size[0]), float(input_size[1]), sigma_t, lambdas_dev,
                             bool(use_target_weight))


class _FusionTerms(torch.autograd.Function):
    """The seven loss values of pk_fusion_terms_fwd with the coordinates HANDED IN (None = the maps' own soft-argmax) plus the per-map
    sigma of compute_heatmap_variance; gradients for heatmaps, offsets, variances and the coordinates (fusion_head.py:405-575,637-743)."""

    @staticmethod
    def forward(ctx, hm, off, var, coords, target, weight, gt, in_w, in_h, sigma_t, lambdas, use_target_weight):
        hm, off, var = _chk(hm, name="heatmaps"), _chk(off, name="offsets"), _chk(var, name="variances")
        target, weight, gt = _chk(target), _chk(weight), _chk(gt)
        co = None if coords is None else _chk(coords, name="coords")
        B, K, H, W = hm.shape
        if off.shape != (B, K, 2, H, W) or var.shape != hm.shape or target.shape != hm.shape or weight.numel() != B * K or gt.shape != (B, K, 2) \
                or (co is not None and co.shape != (B, K, 2)):
            raise _lib.PoseKernelError("fusion terms: inconsistent shapes")
        lambdas = _chk(lambdas.float(), name="lambdas")
        ws = torch.empty(loss_ws_floats(B, K), dtype=F32, device=hm.device)
        losses = torch.empty(7, dtype=F32, device=hm.device)
        sigma = torch.empty(B, K, dtype=F32, device=hm.device)
        call("pk_fusion_terms_fwd", hm, off, var, target, weight, gt, co, ws, losses, sigma, B, K, H, W, float(in_w), float(in_h), float(sigma_t),
             lambdas, 1 if use_target_weight else 0, stream_ptr())
        ctx.save_for_backward(hm, target, ws, lambdas)
        ctx.meta = (float(sigma_t), co is not None, tuple(off.shape))
        return losses, sigma

    @staticmethod
    def backward(ctx, g, gsig):
        hm, target, ws, lambdas = ctx.saved_tensors
        sigma_t, ext, off_shape = ctx.meta
        B, K, H, W = hm.shape
        dhm, dvar = torch.empty_like(hm), torch.empty_like(hm)
        doff = torch.empty(off_shape, dtype=F32, device=hm.device)
        dco = torch.empty(B, K, 2, dtype=F32, device=hm.device) if ext else None
        call("pk_fusion_terms_bwd", hm, target, ws, None, None if gsig is None else _chk(gsig.float()), _chk(g.float()), dhm, doff, dvar, dco,
             B, K, H, W, sigma_t, lambdas, stream_ptr())
        return dhm, doff, dvar, dco, None, None, None, None, None, None, None, None


def fusion_terms(lambdas7, hm=None, off=None, var=None, coords=None, target=None, weight=None, gt=None, input_size=(1.0, 1.0),
                 sigma_t=2.0, use_target_weight=True, shape=None):
    """Per-term access to the fused loss kernels for the reference's public loss methods: inputs a method does not take are replaced by
    neutral device tensors (zero maps; variances == sigma_t, whose term is then exactly 0) and masked out by a zero lambda.
    -> (losses (7,), sigma (B,K))."""
    ref = hm if hm is not None else (off if off is not None else coords)
    dev = ref.device
    B, K, H, W = shape if shape is not None else hm.shape
    z = None

    def zeros():
        nonlocal z
        if z is None:
            z = torch.zeros(B, K, H, W, dtype=F32, device=dev)
        return z

    hm = zeros() if hm is None else hm.float()
    off = torch.zeros(B, K, 2, H, W, dtype=F32, device=dev) if off is None else off.float().reshape(B, K, 2, H, W)
    var = torch.full((B, K, H, W), float(sigma_t), dtype=F32, device=dev) if var is None else var.float()
    target = zeros() if target is None else target.float()
    weight = torch.ones(B, K, 1, dtype=F32, device=dev) if weight is None else weight.float()
    gt = torch.zeros(B, K, 2, dtype=F32, device=dev) if gt is None else gt.float()
    lam = torch.as_tensor(lambdas7, dtype=F32, device=dev) if not torch.is_tensor(lambdas7) else lambdas7
    return _FusionTerms.apply(hm, off, var, None if coords is None else coords.float(), target, weight, gt, float(input_size[0]),
                              float(input_size[1]), float(sigma_t), lam, bool(use_target_weight))


class _SoftArgmax(torch.autograd.Function):
    """SoftArgmax2D.forward with gradients (fusion_head.py:24-71): coords through the softmax, scores to the first maximum."""

    @staticmethod
    def forward(ctx, hm):
        hm = _chk(hm, name="heatmaps")
        one = torch.full((1,), 40.0, device=hm.device)          # sigmoid(40) == 1: pure global soft-argmax
        co, sc = softargmax_refine_decode(hm, None, one, None, radius=0)
        ctx.save_for_backward(hm, co, sc)
        return co, sc

    @staticmethod
    def backward(ctx, gco, gsc):
        hm, co, sc = ctx.saved_tensors
        B, K, H, W = hm.shape
        dhm = torch.empty_like(hm)
        call("pk_softargmax_bwd", hm, co, sc, None if gco is None else _chk(gco.float()), None if gsc is None else _chk(gsc.float()), dhm,
             B * K, H, W, stream_ptr())
        return dhm


def softargmax(hm):
    return _SoftArgmax.apply(hm)


def local_gaussian_refine(heatmaps, coords, radius=2):
    hm, c = _chk(heatmaps, name="heatmaps"), _chk(coords, name="coords")
    B, K, H, W = hm.shape
    out = torch.empty_like(c)
    call("pk_local_gaussian_refine", hm, c, out, B * K, H, W, int(radius), stream_ptr())
    return out


class _RowsByMap(torch.autograd.Function):
    """Rows moved through an int32 row map (window_partition = gather, window_reverse = scatter); the backward is the opposite move."""

    @staticmethod
    def forward(ctx, x2d, rowmap, n_out, scatter):
        x2d = x2d.contiguous()
        if not x2d.is_cuda or (x2d.shape[1] * x2d.element_size()) % 4:
            raise _lib.PoseKernelError("rows_by_map: CUDA(HIP) tensor with rows of whole dwords expected")
        out = (torch.zeros if scatter else torch.empty)((n_out, x2d.shape[1]), dtype=x2d.dtype, device=x2d.device)
        call("pk_rows_by_map", x2d, out, rowmap, rowmap.numel(), x2d.shape[1] * x2d.element_size(), 1 if scatter else 0, stream_ptr())
        ctx.save_for_backward(rowmap)
        ctx.meta = (x2d.shape[0], scatter)
        return out

    @staticmethod
    def backward(ctx, g):
        (rowmap,) = ctx.saved_tensors
        n_in, scatter = ctx.meta
        return _RowsByMap.apply(g, rowmap, n_in, not scatter), None, None, None


def rows_by_map(x2d, rowmap, n_out, scatter):
    return _RowsByMap.apply(x2d, rowmap, n_out, scatter)


class _DropPath(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, keep):
        x = _chk(x, name="x")
        out = torch.empty_like(x)
        call("pk_drop_path_f32", x, mask, out, x.shape[0], x.numel() // x.shape[0], float(keep), stream_ptr())
        ctx.save_for_backward(mask)
        ctx.keep = keep
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return _DropPath.apply(g.float().contiguous(), mask, ctx.keep), None, None


def drop_path(x, mask, keep):
    return _DropPath.apply(x, mask, keep)


class _PixelLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, kind):
        pred, target = _chk(pred), _chk(target)
        B, K = pred.shape[:2]
        HW = pred.numel() // (B * K)
        w = None if weight is None else _chk(weight)
        partial = torch.empty(1024, dtype=F32, device=pred.device)
        loss = torch.empty(1, dtype=F32, device=pred.device)
        call("pk_pixel_loss_fwd", pred, target, w, partial, loss, B, K, HW, kind, stream_ptr())
        ctx.save_for_backward(pred, target, w if w is not None else pred.new_empty(0))
        ctx.kind, ctx.has_w = kind, w is not None
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, target, w = ctx.saved_tensors
        B, K = pred.shape[:2]
        dp = torch.empty_like(pred)
        call("pk_pixel_loss_bwd", pred, target, w if ctx.has_w else None, _chk(g.reshape(1).float()), dp, B, K, pred.numel() // (B * K), ctx.kind,
             stream_ptr())
        return dp, None, None, None


def pixel_loss(pred, target, weight, kind):
    return _PixelLoss.apply(pred, target, weight, int(kind))


class _SpatialStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hm):
        hm = _chk(hm)
        B, K, H, W = hm.shape
        mean = torch.empty(B, K, 2, dtype=F32, device=hm.device)
        var = torch.empty(B, K, 2, dtype=F32, device=hm.device)
        call("pk_spatial_stats", hm, mean, var, B * K, H, W, stream_ptr())
        ctx.save_for_backward(hm, mean, var)
        return mean, var

    @staticmethod
    def backward(ctx, gmean, gvar):
        hm, mean, var = ctx.saved_tensors
        B, K, H, W = hm.shape
        gmean = torch.zeros_like(mean) if gmean is None else gmean.contiguous().float()
        gvar = torch.zeros_like(var) if gvar is None else gvar.contiguous().float()
        dhm = torch.empty_like(hm)
        call("pk_spatial_stats_bwd", hm, mean, var, gmean, gvar, dhm, B * K, H, W, stream_ptr())
        return dhm


def spatial_stats(hm):
    """centre of mass and per-axis variance of hm/(sum+1e-8): (B,K,2), (B,K,2); differentiable w.r.t. hm."""
    return _SpatialStats.apply(hm)


def temporal_smooth(coords, kernel):
    coords = _chk(coords)
    T, K, _ = coords.shape
    w = torch.as_tensor(kernel, dtype=torch.float64).to(coords.device)
    out = torch.empty_like(coords)
    call("pk_temporal_smooth", coords, out, w, T, K * 2, int(w.numel()), stream_ptr())
    return out


def nms_pose(preds, maxvals, distance_threshold=5.0):
    preds, maxvals = _chk(preds), _chk(maxvals.reshape(preds.shape[0], preds.shape[1]))
    B, K, _ = preds.shape
    out = torch.empty_like(preds)
    keep = torch.empty(B, K, dtype=torch.uint8, device=preds.device)
    _call_if(B * K, "pk_nms_pose", preds, maxvals, out, keep, B, K, float(distance_threshold), stream_ptr())
    return out, keep.bool().view(B, K, 1)


# ------------------------------------------------------------------------------------------------ optimiser
def adamw_step(param, grad, exp_avg, exp_avg_sq, flags, param_bf16, lr_dev, step_dev, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    call("pk_adamw_step", param, grad, exp_avg, exp_avg_sq, flags, param_bf16, param.numel(), lr_dev, step_dev, float(beta1), float(beta2),
         float(eps), float(weight_decay), float(grad_scale), stream_ptr())
