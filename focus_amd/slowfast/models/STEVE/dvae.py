"""Discrete VAE of STEVE (mirror of slowfast/models/STEVE/dvae.py:3-32): 4x4/stride-4 patch encoder to vocab logits,
two PixelShuffle(2) stages back to pixels.  Plain nn.Conv2d stacks (ATen/MIOpen) with the reference's module
names, so its checkpoints load unchanged."""
import torch.nn as nn

from .utils import Conv2dBlock, conv2d


class dVAE(nn.Module):
    def __init__(self, vocab_size, img_channels):
        super().__init__()
        self.encoder = nn.Sequential(
            Conv2dBlock(img_channels, 64, 4, 4),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            conv2d(64, vocab_size, 1),
        )
        self.decoder = nn.Sequential(
            Conv2dBlock(vocab_size, 64, 1),
            Conv2dBlock(64, 64, 3, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64 * 2 * 2, 1),
            nn.PixelShuffle(2),
            Conv2dBlock(64, 64, 3, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64, 1, 1),
            Conv2dBlock(64, 64 * 2 * 2, 1),
            nn.PixelShuffle(2),
            conv2d(64, img_channels, 1),
        )
