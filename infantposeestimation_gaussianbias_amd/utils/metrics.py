"""COCOEvaluator / AverageMeter / MetricLogger with the reference's surface (utils/metrics.py:15-325).

`COCOEvaluator.update` takes what `validate()` has at hand.  When the predictions are device tensors the (B,K,3) records and the
per-instance scores come from one kernel (pk_pose_records) and ONE device->host copy per batch; numpy inputs (the reference's
calling convention) are accepted as they are -- the evaluator itself is host bookkeeping (a list of dicts for pycocotools).
COCO AP through pycocotools is third-party code outside the path: `evaluate()` uses it when it is importable and
an annotation file is given, and offers the reference's own OKS matching (`gt_annotations=`) otherwise."""
from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np


class COCOEvaluator:
    DEFAULT_OKS_SIGMAS = np.array([0.026, 0.025, 0.025, 0.035, 0.035, 0.079, 0.079, 0.072, 0.072, 0.062, 0.062, 0.107, 0.107, 0.087,
                                   0.087, 0.089, 0.089])

    def __init__(self, ann_file: Optional[str] = None, oks_sigmas: Optional[np.ndarray] = None, num_keypoints: int = 17):
        self.ann_file = ann_file
        self.oks_sigmas = oks_sigmas if oks_sigmas is not None else self.DEFAULT_OKS_SIGMAS
        self.num_keypoints = num_keypoints
        self.oks_thresholds = np.linspace(0.5, 0.95, 10)
        self.predictions: List[Dict] = []

    def reset(self):
        self.predictions = []

    def update(self, pred_keypoints, pred_scores, image_ids, ann_ids, centers=None, scales=None, areas=None, bboxes=None):
        """pred_keypoints (B,K,2) in image space, pred_scores (B,K): device tensors or numpy arrays; ids / areas / bboxes: sequences,
        arrays or tensors (utils/metrics.py:61-106: centers and scales are accepted and unused there too)."""
        import torch
        if torch.is_tensor(pred_keypoints) and pred_keypoints.is_cuda:
            from .. import hipops
            rec, inst = hipops.pose_records(pred_keypoints.float(), pred_scores.float())
            rec, inst = rec.cpu().numpy().astype(np.float64), inst.cpu().numpy()
        else:
            pk, ps = np.asarray(pred_keypoints), np.asarray(pred_scores)
            rec = np.zeros((pk.shape[0], self.num_keypoints, 3))
            rec[:, :, :2], rec[:, :, 2] = pk, ps
            inst = np.array([ps[i][ps[i] > 0].mean() if (ps[i] > 0).any() else 0.0 for i in range(pk.shape[0])])
        to_list = lambda v: v.tolist() if hasattr(v, "tolist") else list(v)
        image_ids, ann_ids, areas = to_list(image_ids), to_list(ann_ids), to_list(areas)
        bboxes = bboxes.cpu().numpy() if torch.is_tensor(bboxes) else np.asarray(bboxes)
        for i in range(rec.shape[0]):
            self.predictions.append({'image_id': int(image_ids[i]), 'ann_id': int(ann_ids[i]), 'keypoints': rec[i].flatten().tolist(),
                                     'score': float(inst[i]), 'area': float(areas[i]), 'bbox': bboxes[i].tolist()})

    def compute_oks(self, pred_kpts, gt_kpts, gt_vis, area) -> float:
        d = (pred_kpts[:, 0] - gt_kpts[:, 0]) ** 2 + (pred_kpts[:, 1] - gt_kpts[:, 1]) ** 2
        e = d / (2 * area * (self.oks_sigmas ** 2) + np.spacing(1))
        valid = gt_vis > 0
        if valid.sum() == 0:
            return 0.0
        return np.sum(np.exp(-e[valid])) / valid.sum()

    def evaluate(self, gt_annotations: Optional[List[Dict]] = None) -> Dict[str, float]:
        if len(self.predictions) == 0:
            return {'AP': 0.0, 'AP50': 0.0, 'AP75': 0.0}
        if self.ann_file is not None:
            import json
            import os
            import tempfile
            from pycocotools.coco import COCO              # third-party: ImportError here means "install pycocotools", as in the reference
            from pycocotools.cocoeval import COCOeval
            with tempfile.NamedTemporaryFile(mode='w', suffix='.json', delete=False) as f:
                json.dump(self.predictions, f)
                pred_file = f.name
            try:
                coco_gt = COCO(self.ann_file)
                coco_eval = COCOeval(coco_gt, coco_gt.loadRes(pred_file), 'keypoints')
                coco_eval.evaluate()
                coco_eval.accumulate()
                coco_eval.summarize()
                names = ('AP', 'AP50', 'AP75', 'AP_M', 'AP_L', 'AR', 'AR50', 'AR75', 'AR_M', 'AR_L')
                return {n: coco_eval.stats[i] for i, n in enumerate(names)}
            finally:
                os.unlink(pred_file)
        if gt_annotations is not None:
            return self._manual_evaluate(gt_annotations)
        raise ValueError("Either ann_file or gt_annotations must be provided")

    def _manual_evaluate(self, gt_annotations: List[Dict]) -> Dict[str, float]:
        """Greedy OKS matching per image and threshold; "AP" = precision at the threshold (utils/metrics.py:206-270)."""
        pred_by_img, gt_by_img = defaultdict(list), defaultdict(list)
        for pred in self.predictions:
            pred_by_img[pred['image_id']].append(pred)
        for gt in gt_annotations:
            gt_by_img[gt['image_id']].append(gt)
        aps = []
        for thresh in self.oks_thresholds:
            tp = fp = 0
            for img_id, gts in gt_by_img.items():
                matched = set()
                for pred in sorted(pred_by_img[img_id], key=lambda x: x['score'], reverse=True):
                    pk = np.array(pred['keypoints']).reshape(-1, 3)
                    best_oks, best_idx = 0, -1
                    for gi, gt in enumerate(gts):
                        if gi in matched:
                            continue
                        gk = np.array(gt['keypoints']).reshape(-1, 3)
                        oks = self.compute_oks(pk[:, :2], gk[:, :2], gk[:, 2], gt['area'])
                        if oks > best_oks:
                            best_oks, best_idx = oks, gi
                    if best_oks >= thresh and best_idx >= 0:
                        tp += 1
                        matched.add(best_idx)
                    else:
                        fp += 1
            aps.append(tp / (tp + fp + 1e-10))
        return {'AP': np.mean(aps), 'AP50': aps[0] if len(aps) > 0 else 0.0, 'AP75': aps[5] if len(aps) > 5 else 0.0}


class AverageMeter:
    def __init__(self, name: str = '', fmt: str = ':f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val: float, n: int = 1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ('{name} {val' + self.fmt + '} ({avg' + self.fmt + '})').format(**self.__dict__)


class MetricLogger:
    def __init__(self, delimiter: str = '  '):
        self.meters, self.delimiter = defaultdict(AverageMeter), delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            self.meters[k].update(float(v))

    def __str__(self):
        return self.delimiter.join(f'{k}: {m.avg:.4f}' for k, m in self.meters.items())
