"""Utils module."""
from .metrics import AverageMeter, COCOEvaluator, MetricLogger
from . import postprocess

__all__ = ['AverageMeter', 'COCOEvaluator', 'MetricLogger', 'postprocess']
