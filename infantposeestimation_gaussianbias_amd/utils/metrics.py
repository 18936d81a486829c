"""AverageMeter / MetricLogger with the reference's semantics (utils/metrics.py:275-325). COCO AP (pycocotools) is
third-party host code outside the hot path."""
from collections import defaultdict


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
