"""PatchEmbed (mirror of slowfast/models/stem_helper.py:290-320): Conv3d with kernel == stride (Motionformer), executed as
im2col (HIP) + one MFMA GEMM; the parameter keeps the Conv3d shape so reference checkpoints load.  The overlapping,
padded stem of MViT (3x7x7 stride 2x4x4) is a convolution proper and stays with ATen / MIOpen in fp32."""
import torch.nn as nn
import torch.nn.functional as F

from focus_amd import ops


class PatchEmbed(nn.Module):
    def __init__(self, dim_in=3, dim_out=768, kernel=(1, 16, 16), stride=(1, 4, 4), padding=(1, 7, 7), conv_2d=False):
        super().__init__()
        if conv_2d:
            raise NotImplementedError("MVIT.PATCH_2D (image models) is not part of the video path")
        pads = list(padding) if hasattr(padding, "__iter__") else [padding] * 3
        self.patches = list(kernel) == list(stride) and all(p == 0 for p in pads)    # (video_model_builder.py:1134-1141)
        self.kernel = tuple(kernel)
        self.proj = nn.Conv3d(dim_in, dim_out, kernel_size=kernel, stride=stride, padding=padding)
        self.compute_dtype = None     # set by the owning model (fp32 or bf16)

    def forward(self, x):
        if not self.patches:
            y = F.conv3d(x.float(), self.proj.weight, self.proj.bias, self.proj.stride, self.proj.padding)
            return y.flatten(2).transpose(1, 2).to(self.compute_dtype or x.dtype).contiguous()
        kt, kh, kw = self.kernel
        B, Cin, T, H, W = x.shape
        cols = ops.im2col_patches(x, kt, kh, kw, self.compute_dtype or x.dtype)
        w = self.proj.weight.view(self.proj.weight.shape[0], -1)     # [D, Cin*kt*kh*kw]
        y = ops.linear(cols, w, self.proj.bias)
        return y.view(B, (T // kt) * (H // kh) * (W // kw), -1)
