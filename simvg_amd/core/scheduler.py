"""LR scheduler registry -- mirror of the reference's `simvg/core/scheduler.py:1-78`.  Schedulers are stepped once
per EPOCH (tools/train.py:177).  `MultiStepLRWarmUp` is a LambdaLR whose factor for epoch e (0-based) is
(e+1)/(warmup_epochs+1) while e <= warmup_epochs-1, afterwards decay_ratio^(#decay_steps s with e+1 >= s), or -- when
both decay_steps and decay_ratio are None -- a linear ramp down to 0 at max_epoch.  Pinned against the imported
reference class by `tests/golden/apis_golden.pt` (`oracle/make_golden_apis.py`)."""
from collections.abc import Sequence

import torch.optim.lr_scheduler as lr_scheduler

from ..models.builder import Registry

SCHEDULERS = Registry("SCHEDULERS")


def build_scheduler(cfg, optimizer):
    return SCHEDULERS.build(cfg, default_args=dict(optimizer=optimizer))


def multistep_warmup_factor(epoch, warmup_epochs, decay_steps=None, decay_ratio=None, max_epoch=-1):
    if epoch <= warmup_epochs - 1:
        return float(epoch + 1) / float(warmup_epochs + 1)
    if isinstance(decay_steps, Sequence) and decay_ratio > 0.0:
        factor = 1.0
        for step in decay_steps:
            if epoch + 1 < step:
                break
            factor *= decay_ratio
        return factor
    if decay_steps is None and decay_ratio is None:
        span = max_epoch - warmup_epochs
        return (span - (epoch - warmup_epochs)) / span
    raise ValueError("MultiStepLRWarmUp: give both decay_steps (a sequence) and decay_ratio (> 0), or neither")


def _no_verbose(kwargs, verbose):
    # torch >= 2.2 deprecates (and later removes) `verbose`; the reference passes verbose=False, i.e. the default
    if verbose:
        kwargs["verbose"] = verbose
    return kwargs


@SCHEDULERS.register_module()
class MultiStepLRWarmUp(lr_scheduler.LambdaLR):
    def __init__(self, optimizer, warmup_epochs, decay_steps=None, decay_ratio=None, max_epoch=-1, verbose=False):
        assert max_epoch > 0
        super().__init__(optimizer, **_no_verbose(dict(
            lr_lambda=lambda e: multistep_warmup_factor(e, warmup_epochs, decay_steps, decay_ratio, max_epoch)), verbose))


@SCHEDULERS.register_module()
class CosineAnnealingLR(lr_scheduler.CosineAnnealingLR):
    def __init__(self, optimizer, T_max, max_epoch=-1, eta_min=0, verbose=False):
        super().__init__(optimizer, **_no_verbose(dict(T_max=T_max, eta_min=eta_min), verbose))


@SCHEDULERS.register_module()
class CosineAnnealingLRWarmRestarts(lr_scheduler.CosineAnnealingWarmRestarts):
    def __init__(self, optimizer, T_0, max_epoch=-1, T_mult=1, eta_min=0, verbose=False):
        super().__init__(optimizer, T_0, **_no_verbose(dict(T_mult=T_mult, eta_min=eta_min), verbose))
