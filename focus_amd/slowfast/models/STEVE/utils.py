"""Parameter factories, schedules and the Gumbel-softmax of the slot models (mirror of slowfast/models/STEVE/utils.py)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def cosine_anneal(step, start_value, final_value, start_step, final_step):
    """utils.py:8-25 (also slowfast/utils/lr_policy.py:8-23)."""
    assert start_value >= final_value
    assert start_step <= final_step
    if step < start_step:
        return start_value
    if step >= final_step:
        return final_value
    a = 0.5 * (start_value - final_value)
    b = 0.5 * (start_value + final_value)
    progress = (step - start_step) / (final_step - start_step)
    return a * math.cos(math.pi * progress) + b


def linear_warmup(step, start_value, final_value, start_step, final_step):
    """utils.py:28-44."""
    assert start_value <= final_value
    assert start_step <= final_step
    if step < start_step:
        return start_value
    if step >= final_step:
        return final_value
    a = final_value - start_value
    b = start_value
    progress = (step + 1 - start_step) / (final_step - start_step)
    return a * progress + b


def gumbel_softmax(logits, tau=1.0, hard=False, dim=-1, noise=None):
    """utils.py:47-61.  `noise` is the Exp(1) draw the reference makes with torch.empty_like(logits).exponential_();
    drawn here with the same call when not supplied, accepted as an argument so parity tests can fix it."""
    eps = torch.finfo(logits.dtype).tiny
    if noise is None:
        noise = torch.empty_like(logits).exponential_()
    gumbels = -(noise + eps).log()
    gumbels = (logits + gumbels) / tau
    y_soft = F.softmax(gumbels, dim)
    if hard:
        index = y_soft.argmax(dim, keepdim=True)
        y_hard = torch.zeros_like(logits).scatter_(dim, index, 1.0)
        return y_hard - y_soft.detach() + y_soft
    return y_soft


def conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
           padding_mode="zeros", weight_init="xavier"):
    """utils.py:64-79."""
    m = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
    if weight_init == "kaiming":
        nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
    else:
        nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.zeros_(m.bias)
    return m


class Conv2dBlock(nn.Module):
    """conv + ReLU (utils.py:82-92).  Convolutions stay on ATen/MIOpen: the dVAE and the CNN encoder are not on the
    hot path (out of scope for kernels)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.m = conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True, weight_init="kaiming")

    def forward(self, x):
        return F.relu(self.m(x))


def linear(in_features, out_features, bias=True, weight_init="xavier", gain=1.0):
    m = nn.Linear(in_features, out_features, bias)
    if weight_init == "kaiming":
        nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
    else:
        nn.init.xavier_uniform_(m.weight, gain)
    if bias:
        nn.init.zeros_(m.bias)
    return m


def gru_cell(input_size, hidden_size, bias=True):
    m = nn.GRUCell(input_size, hidden_size, bias)
    nn.init.xavier_uniform_(m.weight_ih)
    nn.init.orthogonal_(m.weight_hh)
    if bias:
        nn.init.zeros_(m.bias_ih)
        nn.init.zeros_(m.bias_hh)
    return m
