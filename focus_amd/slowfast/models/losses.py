"""Loss lookup (mirror of slowfast/models/losses.py:40-59, 97-121) for the non-EK hot-path configs."""
from functools import partial

import torch.nn as nn

from focus_amd import ops


class LabelSmoothingCrossEntropy(nn.Module):
    def __init__(self, reduction="mean", smoothing=0.1):
        super().__init__()
        assert smoothing < 1.0
        self.smoothing = smoothing

    def forward(self, x, target):
        return ops.label_smoothing_ce(x, target, self.smoothing)


_LOSSES = {"cross_entropy": partial(LabelSmoothingCrossEntropy, smoothing=0.0),
           "label_smoothing_cross_entropy": LabelSmoothingCrossEntropy}


def get_loss_func(cfg, state="train"):
    name = cfg.MODEL.LOSS_FUNC
    if name not in _LOSSES:
        raise NotImplementedError("Loss {} is not supported".format(name))
    ret = _LOSSES[name]
    if name == "label_smoothing_cross_entropy":
        ret = partial(ret, smoothing=cfg.MIXUP.LABEL_SMOOTH_VALUE)
    return ret
