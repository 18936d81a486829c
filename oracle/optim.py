"""Oracle (test infrastructure only): optimiser grouping, LR schedule and AdamW arithmetic (S1).

Follows train.py:55-128 (`build_optimizer`, `build_scheduler`) and torch.optim.AdamW's
published update.  Pinned by tests/golden/meta.json["schedule"] and model_level.npz (cfg1 trajectory).
"""
import torch


def is_no_decay(name: str) -> bool:
    """train.py:66: substring test on the parameter *name* (so `relative_position_bias_table`
    and `downsample.1.bias` are no-decay, while BN scales inside `fuse_layers.*` / `downsample.1.weight` decay)."""
    return "bias" in name or "bn" in name or "norm" in name


def lr_factor(it, iters_per_epoch, warmup_epochs=5, warmup_lr=5e-7, lr=5e-4, milestones=(170, 200), gamma=0.1):
    """LambdaLR factor at iteration `it` (stepped per iteration, train.py:109-124)."""
    warm = warmup_epochs * iters_per_epoch
    if it < warm:
        r = warmup_lr / lr
        return r + (1 - r) * it / warm
    f = 1.0
    for m in milestones:
        if it >= m * iters_per_epoch:
            f *= gamma
    return f


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """One decoupled-weight-decay Adam update on tensors (in place). `step` counts from 1."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p
