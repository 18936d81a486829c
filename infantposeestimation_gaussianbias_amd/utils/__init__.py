"""Utils module."""
from .metrics import AverageMeter, MetricLogger
from . import postprocess

__all__ = ['AverageMeter', 'MetricLogger', 'postprocess']
