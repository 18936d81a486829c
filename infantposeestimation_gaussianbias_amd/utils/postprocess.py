"""Post-processing (drop-in for the reference's utils/postprocess.py hot-path functions), all on device:
the reference's Python B x K loops with .item() become one kernel launch each."""
import torch

from .. import hipops


def get_max_preds(batch_heatmaps):
    """-> preds (B,K,2) float32 (x,y), maxvals (B,K,1) (postprocess.py:10-34)."""
    _, mv, co = hipops.argmax_decode(batch_heatmaps.float(), 0)
    return co, mv.unsqueeze(-1)


def get_max_preds_with_subpixel(batch_heatmaps):
    """Argmax + Taylor sub-pixel step (postprocess.py:37-75)."""
    _, mv, co = hipops.argmax_decode(batch_heatmaps.float(), 2)
    return co, mv.unsqueeze(-1)


def fused_decode(heatmaps, regression_coords=None, centers=None, scales=None, alpha=0.5):
    """postprocess.py:78-135 incl. its quirks (hard-coded 256 image size; the alpha blend is overwritten by the
    confidence-adaptive blend, so `alpha` has no effect — kept for signature parity)."""
    hp, mv = get_max_preds_with_subpixel(heatmaps)
    H, W = heatmaps.shape[2:]
    sx, sy = (256 / W, 256 / H) if (centers is not None and scales is not None) else (1.0, 1.0)
    reg_scale = 1.0
    if regression_coords is not None:
        regression_coords = regression_coords.float()
        if float(regression_coords.max()) <= 1.0:     # one host sync, exactly where the reference has one
            reg_scale = 256.0
    return hipops.fused_blend(hp, mv.squeeze(-1) if regression_coords is not None else None, regression_coords, sx, sy, reg_scale), mv


def coordinate_refinement(heatmaps, initial_coords, window_size=5):
    return hipops.window_refine(heatmaps.float(), initial_coords.float(), window_size)


def filter_low_confidence(preds, maxvals, threshold=0.3):
    mask = (maxvals > threshold).float()
    return preds * mask, mask


def temporal_smoothing(coords_sequence, window_size=5, method="gaussian"):
    """Reference signature (utils/postprocess.py:187-223): (T,K,2) trajectories smoothed along T on the device (one launch
    instead of 2K numpy round trips).  The weights are built exactly as the reference builds them (float64, one-sided
    'gaussian' or uniform); even windows raise like the reference's shape mismatch does."""
    import numpy as np
    if method == "gaussian":
        sigma = window_size / 3.0
        kernel = np.exp(-np.arange(window_size) ** 2 / (2 * sigma ** 2))
        kernel = kernel / kernel.sum()
    else:
        kernel = np.ones(window_size) / window_size
    return hipops.temporal_smooth(coords_sequence, kernel)


def nms_pose(preds, maxvals, distance_threshold=5.0):
    """Reference signature (utils/postprocess.py:241-267): -> (preds * keep, keep (B,K,1) bool)."""
    return hipops.nms_pose(preds, maxvals, distance_threshold)


def transform_preds(coords, center, scale, output_size, input_size=[256, 256]):
    return hipops.affine_coords(coords.float(), center.float().to(coords.device), scale.float().to(coords.device),
                                1.0 / input_size[0], 1.0 / input_size[1])


def postprocess_predictions(outputs, batch_meta, config):
    heatmaps = outputs['heatmaps']
    alpha = config.TEST.FUSION_ALPHA if hasattr(config.TEST, 'FUSION_ALPHA') else 0.5
    preds, maxvals = fused_decode(heatmaps, outputs.get('coords', None), batch_meta.get('center'), batch_meta.get('scale'), alpha)
    preds = coordinate_refinement(heatmaps, preds)
    preds, mask = filter_low_confidence(preds, maxvals, threshold=0.3)
    if 'center' in batch_meta and 'scale' in batch_meta:
        preds = transform_preds(preds, batch_meta['center'], batch_meta['scale'], output_size=[640, 480])
    return {'preds': preds, 'maxvals': maxvals, 'mask': mask}


def heatmap_to_image_coords(pred_keypoints, center, scale, input_size, heatmap_size):
    """train.py:286-303 / validate.py:100-117 as one launch: heat-px -> input-px -> original image (no Python B x K loop)."""
    mul_x = (input_size[0] / heatmap_size[0]) / input_size[0]
    mul_y = (input_size[1] / heatmap_size[1]) / input_size[1]
    return hipops.affine_coords(pred_keypoints.float(), center.float().to(pred_keypoints.device),
                                scale.float().to(pred_keypoints.device), mul_x, mul_y)
