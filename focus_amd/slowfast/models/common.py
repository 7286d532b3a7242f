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


class DropPath(nn.Module):
    """Stochastic depth (common.py:46-70): a whole sample's residual branch is zeroed with probability drop_prob and the
    survivors are rescaled by 1 / keep.  The blocks of the hot path do not call this module -- they read `drop_prob`
    and fold the draw into the residual add (ops.residual_drop_path, one kernel); forward() is the standalone form
    for any other caller, using the reference's draw (floor(keep + U[0,1)) per sample, one torch.rand of shape
    [B, 1, ...])."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob or 0.0, self.training)


def drop_path(x, drop_prob: float = 0.0, training: bool = False):
    if not training or drop_prob == 0.0:
        return x
    keep = 1 - drop_prob
    per_sample = [x.shape[0]] + [1] * (x.dim() - 1)
    survives = torch.rand(per_sample, dtype=x.dtype, device=x.device).add_(keep).floor_()
    return x.div(keep) * survives
