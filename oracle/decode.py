"""Oracle (test infrastructure only): heatmap decoders, numpy float32.

D1 fusion_head.py:24-172,309-365 · D2 pose_estimator.py:331-373 · D3 utils/postprocess.py:10-336
D4 flip-test merge pose_estimator.py:303-319 · eval glue train.py:328-336.
Pinned by tests/golden/decode.npz.
"""
import numpy as np

F32 = np.float32


def flat_argmax(hm):
    """First-maximum flat index per map (torch.max semantics) and the max value."""
    B, K, H, W = hm.shape
    flat = hm.reshape(B, K, H * W)
    idx = np.argmax(flat, axis=2)  # numpy argmax also returns the first maximum
    return idx.astype(np.int64), np.take_along_axis(flat, idx[..., None], 2)[..., 0]


def soft_argmax(hm, beta=1.0):
    """(B,K,2) expectation of (x,y) under softmax_hw(beta*hm); scores = raw max."""
    B, K, H, W = hm.shape
    z = (hm * F32(beta)).reshape(B, K, -1).astype(np.float64)
    z = z - z.max(-1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(-1, keepdims=True)
    p = p.reshape(B, K, H, W)
    cx = (p * np.arange(W)[None, None, None, :]).sum((2, 3))
    cy = (p * np.arange(H)[None, None, :, None]).sum((2, 3))
    return np.stack([cx, cy], -1).astype(F32), hm.reshape(B, K, -1).max(-1)


def bilinear_border(plane, x, y):
    """grid_sample(mode=bilinear, padding_mode=border, align_corners=True) at pixel (x,y)."""
    H, W = plane.shape
    x = min(max(float(x), 0.0), W - 1.0)
    y = min(max(float(y), 0.0), H - 1.0)
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    fx, fy = x - x0, y - y0
    acc = 0.0
    for yy, wy in ((y0, 1 - fy), (y0 + 1, fy)):
        for xx, wx in ((x0, 1 - fx), (x0 + 1, fx)):
            if 0 <= xx < W and 0 <= yy < H:
                acc += float(plane[yy, xx]) * wx * wy
    return acc


def local_softmax_centroid(hm, coarse, radius=2):
    """fusion_head.py:84-128: softmax-weighted centroid of the clipped (2r+1)² patch around round(coarse)."""
    B, K, H, W = hm.shape
    out = coarse.astype(F32).copy()
    for b in range(B):
        for k in range(K):
            px = int(min(max(np.rint(coarse[b, k, 0]), 0), W - 1))   # rint = round-half-even, as torch.round
            py = int(min(max(np.rint(coarse[b, k, 1]), 0), H - 1))
            x0, x1, y0, y1 = max(0, px - radius), min(W, px + radius + 1), max(0, py - radius), min(H, py + radius + 1)
            patch = hm[b, k, y0:y1, x0:x1].astype(np.float64)
            w = np.exp(patch - patch.max())
            w /= w.sum()
            out[b, k, 0] = (w * np.arange(x0, x1)[None, :]).sum()
            out[b, k, 1] = (w * np.arange(y0, y1)[:, None]).sum()
    return out


def fusion_decode(hm, offsets, alpha_param, fusion_weight_sigmoid, apply_offset=True):
    """D1 `HeatmapRegressionHead.decode`: blend global/local, then add sampled offsets."""
    B, K, H, W = hm.shape
    glob, scores = soft_argmax(hm)
    loc = local_softmax_centroid(hm, glob)
    a = 1.0 / (1.0 + np.exp(-float(alpha_param)))
    coords = (a * glob.astype(np.float64) + (1 - a) * loc.astype(np.float64)).astype(F32)
    if apply_offset:
        res = coords.astype(np.float64).copy()
        for b in range(B):
            for k in range(K):
                for c in range(2):
                    res[b, k, c] += float(fusion_weight_sigmoid) * bilinear_border(offsets[b, k, c], coords[b, k, 0], coords[b, k, 1])
        coords = res.astype(F32)
    return coords, scores


def argmax_decode(hm, shift=True):
    """D2 `PoseEstimator.decode_heatmaps`: argmax + quarter-pixel shift toward the higher neighbour."""
    B, K, H, W = hm.shape
    idx, mx = flat_argmax(hm)
    kp = np.stack([(idx % W), (idx // W)], -1).astype(F32)
    if shift:
        for b in range(B):
            for k in range(K):
                x, y = int(kp[b, k, 0]), int(kp[b, k, 1])
                if 0 < x < W - 1 and 0 < y < H - 1:
                    kp[b, k, 0] += F32(0.25) * np.sign(hm[b, k, y, x + 1] - hm[b, k, y, x - 1])
                    kp[b, k, 1] += F32(0.25) * np.sign(hm[b, k, y + 1, x] - hm[b, k, y - 1, x])
    return kp, mx


def max_preds(hm):
    """D3 `get_max_preds`: float (x,y) of the argmax and maxvals (B,K,1)."""
    B, K, H, W = hm.shape
    idx, mx = flat_argmax(hm)
    return np.stack([idx % W, idx // W], -1).astype(F32), mx[..., None]


def max_preds_taylor(hm):
    """D3 `get_max_preds_with_subpixel` (postprocess.py:37-75); strict 1<px<W-1 bounds."""
    B, K, H, W = hm.shape
    p, mv = max_preds(hm)
    for b in range(B):
        for k in range(K):
            m = hm[b, k]
            x, y = int(p[b, k, 0]), int(p[b, k, 1])
            if 1 < x < W - 1 and 1 < y < H - 1:
                dx = float(F32(m[y, x + 1] - m[y, x - 1]))
                dy = float(F32(m[y + 1, x] - m[y - 1, x]))
                dxx = float(F32(F32(m[y, x + 1] - F32(2) * m[y, x]) + m[y, x - 1]))
                dyy = float(F32(F32(m[y + 1, x] - F32(2) * m[y, x]) + m[y - 1, x]))
                if dxx < 0:
                    p[b, k, 0] += F32(np.clip(dx / (2 * abs(dxx)), -0.5, 0.5))
                if dyy < 0:
                    p[b, k, 1] += F32(np.clip(dy / (2 * abs(dyy)), -0.5, 0.5))
    return p, mv


def fused_decode(hm, regression=None, centers=None, scales=None, alpha=0.5):
    """D3 `fused_decode` (postprocess.py:78-135), including its quirks: hard-coded 256 image
    size and the alpha blend being overwritten by the confidence-adaptive blend."""
    B, K, H, W = hm.shape
    hp, mv = max_preds_taylor(hm)
    if centers is not None and scales is not None:
        hp[:, :, 0] *= F32(256 / W)
        hp[:, :, 1] *= F32(256 / H)
    if regression is None:
        return hp, mv
    reg = np.asarray(regression, F32)
    if reg.max() <= 1.0:
        reg = reg * F32(256)
    a = mv / (mv + F32(0.1))
    return (a * hp + (1 - a) * reg).astype(F32), mv


def window_refine(hm, coords, window=5):
    """D3 `coordinate_refinement` (postprocess.py:138-184): weights = patch/(sum+1e-8), int() truncation."""
    B, K, H, W = hm.shape
    out = np.asarray(coords, F32).copy()
    hw = window // 2
    for b in range(B):
        for k in range(K):
            x, y = int(coords[b, k, 0]), int(coords[b, k, 1])
            x0, x1, y0, y1 = max(0, x - hw), min(W, x + hw + 1), max(0, y - hw), min(H, y + hw + 1)
            if x1 <= x0 or y1 <= y0:
                continue
            patch = hm[b, k, y0:y1, x0:x1].astype(np.float64)
            w = patch / (patch.sum() + 1e-8)
            out[b, k, 0] = (w.sum(0) * np.arange(x0, x1)).sum()
            out[b, k, 1] = (w.sum(1) * np.arange(y0, y1)).sum()
    return out


def filter_low_confidence(preds, maxvals, threshold=0.3):
    mask = (maxvals > threshold).astype(F32)
    return preds * mask, mask


def transform_preds_batch(coords, center, scale, input_size=(256, 256)):
    """D3 `transform_preds` (postprocess.py:270-292): model space -> crop box in image space."""
    out = np.asarray(coords, F32).copy()
    for c in range(2):
        out[:, :, c] = coords[:, :, c] * (scale[:, None, c] / F32(input_size[c])) + center[:, None, c] - scale[:, None, c] / 2
    return out


def postprocess_pipeline(hm, regression, center, scale, alpha=0.5):
    """D3 `postprocess_predictions` (postprocess.py:296-336)."""
    p, mv = fused_decode(hm, regression, center, scale, alpha)
    p = window_refine(hm, p)
    p, mask = filter_low_confidence(p, mv, 0.3)
    if center is not None and scale is not None:
        p = transform_preds_batch(p, center, scale)
    return p, mv, mask


def eval_transform(coords, center, scale, input_size):
    """train.py:328-336 / validate.py:32-37: coords/input_size*scale + center - scale/2 (per axis)."""
    out = np.asarray(coords, F32).copy()
    for c in range(2):
        out[:, :, c] = coords[:, :, c] / F32(input_size[c]) * scale[:, None, c] + center[:, None, c] - scale[:, None, c] / 2
    return out


def flip_merge(hm, hm_from_flipped, flip_pairs):
    """D4: un-flip along W, swap L/R channels, average (pose_estimator.py:309-319)."""
    back = hm_from_flipped[..., ::-1].copy()
    sw = back.copy()
    for a, b in flip_pairs:
        sw[:, a], sw[:, b] = back[:, b], back[:, a]
    return ((hm + sw) / F32(2)).astype(F32)


def temporal_smoothing(coords, window_size=5, method="gaussian"):
    """utils/postprocess.py:187-223: per joint coordinate np.convolve(edge-padded trajectory, kernel, 'valid'); the 'gaussian'
    kernel is the reference's one-sided exp(-n^2/(2 sigma^2)), n = 0..w-1, sigma = w/3, normalised.  coords (T,K,2) float32."""
    coords = np.asarray(coords, np.float32)
    T, K, _ = coords.shape
    if method == "gaussian":
        sigma = window_size / 3.0
        kernel = np.exp(-np.arange(window_size) ** 2 / (2 * sigma ** 2))
        kernel = kernel / kernel.sum()
    else:
        kernel = np.ones(window_size) / window_size
    half = window_size // 2
    out = coords.copy()
    for k in range(K):
        for d in range(2):
            padded = np.pad(coords[:, k, d], (half, half), mode="edge")
            out[:, k, d] = np.convolve(padded, kernel, mode="valid").astype(np.float32)
    return out


def nms_pose(preds, maxvals, distance_threshold=5.0):
    """utils/postprocess.py:241-267 restated loop for loop.  preds (B,K,2), maxvals (B,K,1) -> (preds*keep, keep (B,K,1) bool)."""
    preds, maxvals = np.asarray(preds, np.float32), np.asarray(maxvals, np.float32)
    B, K, _ = preds.shape
    keep = np.ones((B, K, 1), bool)
    for b in range(B):
        for k in range(K):
            if not keep[b, k, 0]:
                continue
            dist = np.sqrt(((preds[b] - preds[b, k]) ** 2).sum(1, dtype=np.float32))
            nearby = dist < np.float32(distance_threshold)
            idx = np.where(nearby)[0]
            if len(idx) > 1:
                best = idx[int(np.argmax(maxvals[b, idx, 0]))]
                for j in idx:
                    if j != best:
                        keep[b, j, 0] = False
    return preds * keep.astype(np.float32), keep
