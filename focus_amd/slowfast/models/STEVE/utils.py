"""Layer factories and the Gumbel-softmax relaxation of the slot models (reference: slowfast/models/STEVE/utils.py).

Seed-for-seed construction parity with the reference needs the same torch initialisers called in the same order on the
same parameter shapes (weight first, then bias); `_init` is the one place that encodes it.  The step schedules live
in focus_amd/slowfast/utils/lr_policy.py and are re-exported here under the reference's names."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from focus_amd.slowfast.utils.lr_policy import cosine_anneal, linear_warmup  # noqa: F401  (utils.py:8-44)


def _init(module, scheme="xavier", gain=1.0):
    """utils.py:71-78, :100-107: kaiming-uniform (ReLU fan) or xavier-uniform weight, zero bias."""
    if scheme == "kaiming":
        nn.init.kaiming_uniform_(module.weight, nonlinearity="relu")
    elif scheme == "xavier":
        nn.init.xavier_uniform_(module.weight, gain)
    else:
        raise ValueError("unknown weight_init %r" % (scheme,))
    if module.bias is not None:
        nn.init.zeros_(module.bias)
    return module


def conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
           padding_mode="zeros", weight_init="xavier"):
    """nn.Conv2d with the slot models' initialisation (utils.py:64-79)."""
    return _init(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                           groups=groups, bias=bias, padding_mode=padding_mode), weight_init)


def linear(in_features, out_features, bias=True, weight_init="xavier", gain=1.0):
    """nn.Linear with the slot models' initialisation (utils.py:95-108)."""
    return _init(nn.Linear(in_features, out_features, bias=bias), weight_init, gain)


def gru_cell(input_size, hidden_size, bias=True):
    """nn.GRUCell, xavier input weights / orthogonal recurrent weights / zero biases (utils.py:111-118).  The module only
    holds the parameters: the slot update runs them through ops.gru_cell."""
    cell = nn.GRUCell(input_size, hidden_size, bias=bias)
    nn.init.xavier_uniform_(cell.weight_ih)
    nn.init.orthogonal_(cell.weight_hh)
    for b in ((cell.bias_ih, cell.bias_hh) if bias else ()):
        nn.init.zeros_(b)
    return cell


class Conv2dBlock(nn.Module):
    """Convolution (kaiming) followed by ReLU; the submodule is named `m` as in utils.py:82-92.  Convolutions stay on
    ATen/MIOpen: the dVAE and the CNN encoder are callers of the hot path, not kernel targets."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.m = conv2d(in_channels, out_channels, kernel_size, stride, padding, weight_init="kaiming")

    def forward(self, x):
        return self.m(x).relu_()


def gumbel_softmax(logits, tau=1.0, hard=False, dim=-1, noise=None):
    """Relaxed one-hot sample (utils.py:47-61): softmax((logits + G) / tau) with G = -log(E + tiny), E ~ Exp(1).

    `noise` is E.  The reference draws it with torch.empty_like(logits).exponential_(); the same call is made here when
    it is not supplied, and parity tests pass the reference's own draw.  hard=True returns the straight-through one-hot:
    forward value argmax one-hot, gradient of the soft sample."""
    if noise is None:
        noise = torch.empty_like(logits).exponential_()
    tiny = torch.finfo(logits.dtype).tiny
    soft = F.softmax((logits - (noise + tiny).log()) / tau, dim)
    if not hard:
        return soft
    one_hot = torch.zeros_like(logits).scatter_(dim, soft.argmax(dim, keepdim=True), 1.0)
    return one_hot - soft.detach() + soft
