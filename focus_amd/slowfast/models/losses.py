"""Loss lookup (mirror of slowfast/models/losses.py:40-121): label-smoothing / plain cross entropy on the fused
focus_xent_ls kernel, and the EPIC-Kitchens verb+noun wrapper (EKLoss, losses.py:62-95)."""
from functools import partial

import torch.nn as nn

from focus_amd import ops


class LabelSmoothingCrossEntropy(nn.Module):
    def __init__(self, reduction="mean", smoothing=0.1):
        super().__init__()
        assert smoothing < 1.0
        if reduction != "mean":
            raise NotImplementedError("the reference's LabelSmoothingCrossEntropy always returns the mean "
                                      "(losses.py:53-59); reduction=%r is not used by the train loops" % reduction)
        self.smoothing = smoothing
        self.confidence = 1.0 - smoothing

    def forward(self, x, target):
        return ops.label_smoothing_ce(x, target, self.smoothing)


class EKLoss(nn.Module):
    """losses.py:62-95: {'verb_loss', 'noun_loss'} from the two heads' logits and a {'verb','noun'} label dict."""

    def __init__(self, reduction="mean", ce_type="", smoothing=0.1):
        super().__init__()
        self.reduction = reduction
        if ce_type == "soft":
            raise NotImplementedError("soft-target CE (mixup) is outside the hot-path configs (MIXUP.ENABLE False)")
        elif ce_type == "label_smoothing":
            self.ce_loss = LabelSmoothingCrossEntropy(reduction=reduction, smoothing=smoothing)
        else:
            self.ce_loss = LabelSmoothingCrossEntropy(reduction=reduction, smoothing=0.0)   # nn.CrossEntropyLoss

    def forward(self, extra_preds, y):
        return {"verb_loss": self.ce_loss(extra_preds["verb"], y["verb"]),
                "noun_loss": self.ce_loss(extra_preds["noun"], y["noun"])}


_LOSSES = {"cross_entropy": partial(LabelSmoothingCrossEntropy, smoothing=0.0),
           "label_smoothing_cross_entropy": LabelSmoothingCrossEntropy}


def get_loss_func(cfg, state="train"):
    name = cfg.MODEL.LOSS_FUNC
    if state == "val" and name == "soft_cross_entropy":
        name = "cross_entropy"
    if cfg.TRAIN.DATASET == "epickitchens":                       # losses.py:106-114
        if name == "cross_entropy":
            return partial(EKLoss, ce_type="")
        if name == "label_smoothing_cross_entropy":
            return partial(EKLoss, ce_type="label_smoothing", smoothing=cfg.MIXUP.LABEL_SMOOTH_VALUE)
        raise NotImplementedError("%s for epickitchens" % name)
    if name not in _LOSSES:
        raise NotImplementedError("Loss {} is not supported".format(name))
    ret = _LOSSES[name]
    if name == "label_smoothing_cross_entropy":
        ret = partial(ret, smoothing=cfg.MIXUP.LABEL_SMOOTH_VALUE)
    return ret
