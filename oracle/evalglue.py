"""Oracle (test infrastructure only): the evaluator bookkeeping of the reference's validation loop, restated in numpy.

utils/metrics.py:61-106 (`COCOEvaluator.update`: per-instance (K,3) records and the mean of the positive scores), :107-143 (`compute_oks`)
and :206-270 (greedy OKS matching, "AP" = precision at each threshold).  Pinned by tests/golden/extra_r02.npz + meta.json["extra"]["eval"],
captured from the reference's own class.  Never imported by the product package.
"""
import numpy as np

OKS_SIGMAS = np.array([0.026, 0.025, 0.025, 0.035, 0.035, 0.079, 0.079, 0.072, 0.072, 0.062, 0.062, 0.107, 0.107, 0.087, 0.087, 0.089, 0.089])


def records(pred_keypoints, pred_scores, image_ids, ann_ids, areas, bboxes):
    out = []
    for i in range(pred_keypoints.shape[0]):
        rec = np.zeros((pred_keypoints.shape[1], 3))
        rec[:, :2], rec[:, 2] = pred_keypoints[i], pred_scores[i]
        pos = pred_scores[i] > 0
        score = pred_scores[i][pos].mean() if pos.sum() > 0 else 0.0
        out.append({"image_id": int(image_ids[i]), "ann_id": int(ann_ids[i]), "keypoints": rec.flatten().tolist(), "score": float(score),
                    "area": float(areas[i]), "bbox": bboxes[i].tolist()})
    return out


def oks(pred_xy, gt_xy, gt_vis, area, sigmas=OKS_SIGMAS):
    d = ((pred_xy - gt_xy) ** 2).sum(-1)
    e = d / (2 * area * sigmas ** 2 + np.spacing(1))
    valid = gt_vis > 0
    return 0.0 if valid.sum() == 0 else float(np.exp(-e[valid]).sum() / valid.sum())


def precision_metrics(predictions, gts, thresholds=np.linspace(0.5, 0.95, 10)):
    by_img_p, by_img_g = {}, {}
    for p in predictions:
        by_img_p.setdefault(p["image_id"], []).append(p)
    for g in gts:
        by_img_g.setdefault(g["image_id"], []).append(g)
    aps = []
    for th in thresholds:
        tp = fp = 0
        for img, gl in by_img_g.items():
            used = set()
            for p in sorted(by_img_p.get(img, []), key=lambda q: q["score"], reverse=True):
                pk = np.array(p["keypoints"]).reshape(-1, 3)
                best, bi = 0, -1
                for gi, g in enumerate(gl):
                    if gi in used:
                        continue
                    gk = np.array(g["keypoints"]).reshape(-1, 3)
                    v = oks(pk[:, :2], gk[:, :2], gk[:, 2], g["area"])
                    if v > best:
                        best, bi = v, gi
                if best >= th and bi >= 0:
                    tp += 1
                    used.add(bi)
                else:
                    fp += 1
        aps.append(tp / (tp + fp + 1e-10))
    return {"AP": float(np.mean(aps)), "AP50": float(aps[0]), "AP75": float(aps[5])}
