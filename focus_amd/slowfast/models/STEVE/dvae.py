"""Discrete VAE of STEVE (same module tree as slowfast/models/STEVE/dvae.py:3-32, so its checkpoints load unchanged).

The two stacks are written as layer tables: the encoder is a 4x4/stride-4 patch convolution followed by six 1x1
conv+ReLU layers and a 1x1 projection to vocabulary logits; the decoder is two (3x3, 1x1, 1x1, 1x1 -> 4x channels,
PixelShuffle(2)) stages between a 1x1 input and a 1x1 output projection.  The convolutions stay on ATen/MIOpen: the dVAE
is not on the hot path (SURVEY.md section 8 lists it as a caller of the slot update, not a kernel target)."""
import torch.nn as nn

from .utils import Conv2dBlock, conv2d

_WIDTH = 64
_UP = "shuffle"                      # table entry for nn.PixelShuffle(2)
# (out_channels, kernel, stride, padding); input channels chain from the previous entry
_ENCODER = [(_WIDTH, 4, 4, 0)] + [(_WIDTH, 1, 1, 0)] * 6
_UPSAMPLE_STAGE = [(_WIDTH, 3, 1, 1), (_WIDTH, 1, 1, 0), (_WIDTH, 1, 1, 0), (4 * _WIDTH, 1, 1, 0), _UP]
_DECODER = [(_WIDTH, 1, 1, 0)] + _UPSAMPLE_STAGE * 2


def _stack(in_channels, table, out_channels):
    """nn.Sequential of conv+ReLU blocks (and pixel shuffles) from a table, closed by a plain 1x1 projection."""
    layers, width = [], in_channels
    for entry in table:
        if entry == _UP:
            layers.append(nn.PixelShuffle(2))
            width //= 4
            continue
        out, kernel, stride, pad = entry
        layers.append(Conv2dBlock(width, out, kernel, stride, pad))
        width = out
    layers.append(conv2d(width, out_channels, 1))
    return nn.Sequential(*layers)


class dVAE(nn.Module):
    def __init__(self, vocab_size, img_channels):
        super().__init__()
        self.encoder = _stack(img_channels, _ENCODER, vocab_size)
        self.decoder = _stack(vocab_size, _DECODER, img_channels)
