"""Oracle (test infrastructure only): loss functions as pure torch-CPU functions.

L1/L2 FusionPoseLoss + GaussianDistributionConstraint: models/fusion_head.py:372-806.
L3 KeypointMSELoss: models/pose_estimator.py:102-143.  L4: models/losses.py:10-284.
All functions are differentiable through autograd (used to check the hand-derived HIP backward)
and dtype-agnostic (run them in float64 for a tighter reference).  Pinned by
tests/golden/head_loss.npz.
"""
import math

import torch

SKELETON = ((0, 1), (0, 2), (1, 3), (2, 4), (5, 6), (5, 7), (7, 9), (6, 8), (8, 10), (5, 11), (6, 12),
            (11, 12), (11, 13), (13, 15), (12, 14), (14, 16))   # fusion_head.py:389-394
FUSION_WEIGHTS = (1.0, 1.0, 0.5, 0.1, 0.05, 0.05)            # pose_estimator.py:199-206
NAMES = ("heatmap_loss", "offset_loss", "peak_loss", "variance_loss", "overlap_loss", "shape_loss", "total_loss")


def _grids(H, W, like):
    xs = torch.arange(W, dtype=like.dtype).view(1, 1, 1, W)
    ys = torch.arange(H, dtype=like.dtype).view(1, 1, H, 1)
    return xs, ys


def soft_argmax(hm):
    B, K, H, W = hm.shape
    p = torch.softmax(hm.reshape(B, K, -1), -1).reshape(B, K, H, W)
    xs, ys = _grids(H, W, hm)
    return torch.stack([(p * xs).sum((2, 3)), (p * ys).sum((2, 3))], -1), p


def sample_border(planes, cx, cy):
    """Bilinear sample of planes (B,K,C,H,W) at pixel coords (B,K), clamped to the map (== grid_sample
    align_corners=True, padding_mode=border).  Differentiable w.r.t. planes AND the coordinates."""
    B, K, C, H, W = planes.shape
    x = cx.clamp(0, W - 1)
    y = cy.clamp(0, H - 1)
    x0 = x.detach().floor()
    y0 = y.detach().floor()
    fx, fy = x - x0, y - y0
    x0i, y0i = x0.long(), y0.long()
    x1i, y1i = (x0i + 1).clamp(max=W - 1), (y0i + 1).clamp(max=H - 1)   # weight is 0 whenever the clamp bites
    flat = planes.reshape(B, K, C, H * W)

    def at(yi, xi):
        return torch.gather(flat, 3, (yi * W + xi)[:, :, None, None].expand(B, K, C, 1))[..., 0]

    fx, fy = fx[..., None], fy[..., None]
    return at(y0i, x0i) * (1 - fx) * (1 - fy) + at(y0i, x1i) * fx * (1 - fy) + at(y1i, x0i) * (1 - fx) * fy + at(y1i, x1i) * fx * fy


def fusion_pose_loss(hm, offsets, variances, target, weight, gt_keypoints, input_size, sigma_t=2.0,
                     lambdas=FUSION_WEIGHTS, overlap_threshold=0.5, use_target_weight=True):
    """All seven entries of `FusionPoseLoss.forward` (already multiplied by their λ), as a dict.

    hm/target/variances (B,K,H,W), offsets (B,K,2,H,W), weight (B,K,1), gt (B,K,2) input-px,
    input_size (W_in,H_in)."""
    B, K, H, W = hm.shape
    w = weight.reshape(B, K)
    S = w.sum() + 1e-8
    c, P = soft_argmax(hm)
    g = torch.stack([gt_keypoints[..., 0] * (W / input_size[0]), gt_keypoints[..., 1] * (H / input_size[1])], -1)

    sampled = sample_border(offsets, c[..., 0], c[..., 1])
    e_hm, e_off = ((hm - target) ** 2).mean((2, 3)), torch.nn.functional.smooth_l1_loss(sampled, g - c, reduction="none").mean(-1)
    e_peak = ((c - g) ** 2).sum(-1)
    if use_target_weight:
        l_hm, l_off, l_peak = (w * e_hm).sum() / S, (w * e_off).sum() / S, (w * e_peak).sum() / S
    else:       # fusion_head.py:653-657, 708-712, 739-743: plain means over (B, K); the constraint terms stay weighted
        l_hm, l_off, l_peak = e_hm.mean(), e_off.mean(), e_peak.mean()

    xs, ys = _grids(H, W, hm)
    pos = torch.relu(hm)
    Q = pos / (pos.sum((2, 3), keepdim=True) + 1e-8)
    spread = (Q * ((xs - c[..., 0, None, None]) ** 2 + (ys - c[..., 1, None, None]) ** 2)).sum((2, 3))
    sig = torch.sqrt(spread + 1e-8)
    l_var = (w * ((sig - sigma_t) ** 2 + (variances.mean((2, 3)) - sigma_t) ** 2)).sum() / S

    s = torch.sigmoid(hm)
    ssum = s.sum((2, 3))
    num, den = hm.new_zeros(()), hm.new_zeros(())
    for i, j in SKELETON:
        if i >= K or j >= K:
            continue
        ratio = torch.minimum(s[:, i], s[:, j]).sum((1, 2)) / (torch.minimum(ssum[:, i], ssum[:, j]) + 1e-8)
        v = w[:, i] * w[:, j]
        num = num + (torch.relu(ratio - overlap_threshold) * v).sum()
        den = den + v.sum()
    l_ovl = num / (den + 1e-8)

    ent = -(P * torch.log(P + 1e-8)).sum((2, 3))
    l_shape = (w * (ent - math.log(2 * math.pi * math.e * sigma_t ** 2)) ** 2).sum() / S

    parts = [lam * v for lam, v in zip(lambdas, (l_hm, l_off, l_peak, l_var, l_ovl, l_shape))]
    out = dict(zip(NAMES[:6], parts))
    out["total_loss"] = sum(parts)
    return out


def keypoint_mse(pred, target, weight=None):
    """L3: mean over all B*K*HW of (pred*w - target*w)²."""
    B, K = pred.shape[:2]
    p, t = pred.reshape(B, K, -1), target.reshape(B, K, -1)
    if weight is not None:
        p, t = p * weight, t * weight
    return ((p - t) ** 2).mean()


def fused_pose_loss(pred, target, weight=None, kind="mse"):
    """L4 `FusedPoseLoss`: per-pixel loss times w, mean over everything."""
    d = pred - target
    if kind == "mse":
        e = d ** 2
    elif kind == "smoothl1":
        e = torch.where(d.abs() < 1, 0.5 * d ** 2, d.abs() - 0.5)
    else:
        raise ValueError(f"Unsupported loss type: {kind}")
    if weight is not None:
        e = e * weight.reshape(pred.shape[0], pred.shape[1], 1, 1)
    return e.mean()


def spatial_stats(hm):
    """L4 centre of mass (B,K,2) and per-axis variance (B,K,2) of hm/(Σhm+1e-8)."""
    B, K, H, W = hm.shape
    p = hm / (hm.sum((2, 3), keepdim=True) + 1e-8)
    xs, ys = _grids(H, W, hm)
    mx, my = (p * xs).sum((2, 3)), (p * ys).sum((2, 3))
    vx = (p * (xs - mx[..., None, None]) ** 2).sum((2, 3))
    vy = (p * (ys - my[..., None, None]) ** 2).sum((2, 3))
    return torch.stack([mx, my], -1), torch.stack([vx, vy], -1)


def morphology_shape_loss(pred, target, weight=None, lambda_variance=1.0, lambda_mean=0.5):
    pm, pv = spatial_stats(pred)
    tm, tv = spatial_stats(target)
    e = lambda_variance * (pv - tv) ** 2 + lambda_mean * (pm - tm) ** 2
    if weight is not None:
        e = e * weight.reshape(e.shape[0], e.shape[1], 1)
    return e.mean()


def offset_regression_loss(pred, target, weight=None, kind="smoothl1"):
    d = pred - target
    e = {"smoothl1": torch.where(d.abs() < 1, 0.5 * d ** 2, d.abs() - 0.5), "l1": d.abs(), "mse": d ** 2}[kind]
    if weight is not None:
        e = e * weight.reshape(e.shape[0], e.shape[1], 1)
    return e.mean()


def joints_mse_loss(pred, target, weight, use_weight=True):
    """L4 `JointsMSELoss`: Σ_k 0.5*mean_{b,hw}((p_k w_k - t_k w_k)²) / K."""
    B, K = pred.shape[:2]
    p, t = pred.reshape(B, K, -1), target.reshape(B, K, -1)
    if use_weight:
        p, t = p * weight, t * weight
    return (0.5 * ((p - t) ** 2).mean((0, 2))).sum() / K


def combined_loss(pred_hm, pred_coords, pred_refined, tgt_hm, tgt_coords, weight, morph_lambda, morph_weight, reg_weight):
    hm = fused_pose_loss(pred_hm, tgt_hm, weight, "mse")
    mo = morphology_shape_loss(pred_hm, tgt_hm, weight, morph_lambda, 0.5)
    rg = offset_regression_loss(pred_coords, tgt_coords, weight)
    rf = offset_regression_loss(pred_refined, tgt_coords, weight)
    return hm + morph_weight * mo + reg_weight * rg + reg_weight * rf, (hm, mo, rg, rf)
