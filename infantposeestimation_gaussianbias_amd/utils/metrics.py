"""COCOEvaluator / AverageMeter / MetricLogger with the reference's surface (utils/metrics.py:15-325).

`COCOEvaluator.update` takes what `validate()` has at hand.  When the predictions are device tensors the (B,K,3) records and the
per-instance scores come from one kernel (pk_pose_records) and ONE device->host copy per batch; numpy inputs (the reference's
calling convention) are accepted as they are -- the evaluator itself is host bookkeeping (a list of dicts for pycocotools).
COCO AP through pycocotools is third-party code outside the path: `evaluate()` hands it the records when an
annotation file is given, and otherwise reports the precision of greedy OKS matching against `gt_annotations=` (array form)."""
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

    def compute_oks(self, pred_kpts: np.ndarray, gt_kpts: np.ndarray, gt_vis: np.ndarray, area: float) -> float:
        """Object keypoint similarity of one prediction / ground-truth pair (utils/metrics.py:108-142): mean over the visible joints
        of exp(-d^2 / (2 area sigma^2 + eps)); 0 without a visible joint.  Same formula as one cell of `oks_precision`'s matrix."""
        pred_kpts, gt_kpts = np.asarray(pred_kpts, np.float64), np.asarray(gt_kpts, np.float64)
        d2 = ((pred_kpts[:, :2] - gt_kpts[:, :2]) ** 2).sum(-1)
        e = d2 / (2 * area * (np.asarray(self.oks_sigmas, np.float64) ** 2) + np.spacing(1))
        valid = np.asarray(gt_vis) > 0
        return float(np.exp(-e[valid]).sum() / valid.sum()) if valid.any() else 0.0

    def _manual_evaluate(self, gt_annotations: List[Dict]) -> Dict[str, float]:
        """The reference's fallback evaluator (utils/metrics.py:206-270): greedy score-ordered OKS matching per threshold."""
        return oks_precision(self.predictions, gt_annotations, self.oks_sigmas, self.oks_thresholds)

    def evaluate(self, gt_annotations: Optional[List[Dict]] = None) -> Dict[str, float]:
        """AP numbers for the collected records.  With an annotation file the arithmetic is pycocotools' (third-party, outside the
        path: SURVEY §2 row 12 -- this only hands it the records); without one, `gt_annotations` (dicts with image_id / keypoints /
        area) are matched by OKS (`oks_precision`), the stand-in train.py uses to pick best.pth on loaders that carry no file."""
        if not self.predictions:
            return {'AP': 0.0, 'AP50': 0.0, 'AP75': 0.0}
        if self.ann_file is not None:
            return _pycocotools_keypoint_stats(self.ann_file, self.predictions)
        if gt_annotations is None:
            raise ValueError("Either ann_file or gt_annotations must be provided")
        return oks_precision(self.predictions, gt_annotations, self.oks_sigmas, self.oks_thresholds)


def _pycocotools_keypoint_stats(ann_file, records):
    from pycocotools.coco import COCO              # ImportError here means "install pycocotools", as in the reference
    from pycocotools.cocoeval import COCOeval
    gt = COCO(ann_file)
    ev = COCOeval(gt, gt.loadRes(list(records)), 'keypoints')       # loadRes takes the list itself: no temporary json file
    for stage in (ev.evaluate, ev.accumulate, ev.summarize):
        stage()
    return dict(zip(('AP', 'AP50', 'AP75', 'AP_M', 'AP_L', 'AR', 'AR50', 'AR75', 'AR_M', 'AR_L'), ev.stats))


def oks_precision(records, gts, sigmas, thresholds):
    """Precision of score-ordered greedy OKS matching at each threshold (the quantity utils/metrics.py:206-270 reports as "AP").

    Per image ONE (predictions x ground truths) OKS matrix is built with array arithmetic and shared by all thresholds; only the
    greedy assignment (inherently sequential: a matched ground truth leaves the pool) walks it row by row."""
    var2 = 2.0 * np.asarray(sigmas, np.float64) ** 2
    img_of_gt = np.array([g['image_id'] for g in gts])
    img_of_pr = np.array([r['image_id'] for r in records])
    gk = np.array([g['keypoints'] for g in gts], np.float64).reshape(len(gts), -1, 3)
    pk = np.array([r['keypoints'] for r in records], np.float64).reshape(len(records), -1, 3)
    g_area = np.array([g['area'] for g in gts], np.float64)
    p_score = np.array([r['score'] for r in records], np.float64)
    hits = np.zeros(len(thresholds), np.int64)
    tried = 0
    for img in dict.fromkeys(img_of_gt.tolist()):
        gi, pi = np.flatnonzero(img_of_gt == img), np.flatnonzero(img_of_pr == img)
        pi = pi[np.argsort(-p_score[pi], kind="stable")]
        tried += len(pi)
        if not len(pi):
            continue
        d2 = ((pk[pi, None, :, :2] - gk[None, gi, :, :2]) ** 2).sum(-1)                         # (P, G, K)
        vis = gk[gi, :, 2] > 0                                                                      # (G, K)
        e = np.exp(-d2 / (g_area[gi, None] * var2[None, :] + np.spacing(1))[None])
        nvis = vis.sum(-1)
        oks = np.where(nvis > 0, (e * vis[None]).sum(-1) / np.maximum(nvis, 1), 0.0)                # (P, G)
        for t, th in enumerate(thresholds):
            free = np.ones(len(gi), bool)
            for row in oks:
                cand = np.where(free, row, -1.0)
                j = int(cand.argmax())                     # first maximum, like the reference's strict '>' scan
                if cand[j] > 0 and cand[j] >= th:
                    free[j] = False
                    hits[t] += 1
    prec = hits / (tried + 1e-10)
    return {'AP': prec.mean(), 'AP50': prec[0], 'AP75': prec[5] if len(prec) > 5 else 0.0}


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
