"""Hot-loop helpers (mirror of slowfast/utils/misc.py:26-33, 388-398)."""
import math

import torch


def check_nan_losses(loss):
    if math.isnan(loss):
        raise RuntimeError("ERROR: Got NaN losses")


def iter_to_cuda(batch):
    def _to(x):
        if isinstance(x, torch.Tensor):
            return x.cuda(non_blocking=True)
        if isinstance(x, (list, tuple)):
            return type(x)(_to(v) for v in x)
        if isinstance(x, dict):
            return {k: _to(v) for k, v in x.items()}
        return x
    return _to(batch)
