"""Mlp / DropPath (mirror of slowfast/models/common.py:7-70)."""
import torch
import torch.nn as nn

from focus_amd import ops


class Mlp(nn.Module):
    """fc1 -> GELU(erf) -> fc2, one fused autograd op (two MFMA GEMMs with bias/GELU epilogues)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("hot path Mlp uses GELU (common.py:20)")
        if drop > 0.0:
            raise NotImplementedError("MF.DROP > 0 is not used by any hot-path config")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.drop_rate = drop
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, residual=None):
        """residual (optional) is added in the fc2 epilogue: returns residual + mlp(x)."""
        return ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual=residual,
                       act=ops.EPI_GELU)


def drop_path(x, drop_prob: float = 0.0, training: bool = False):
    """Stochastic depth per sample (common.py:46-60)."""
    if drop_prob == 0.0 or not training:
        return x
    keep = 1 - drop_prob
    mask = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
    mask.floor_()
    return x.div(keep) * mask


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)
