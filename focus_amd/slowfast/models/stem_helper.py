"""PatchEmbed (mirror of slowfast/models/stem_helper.py:290-320): Conv3d with kernel == stride, executed as
im2col (HIP) + one MFMA GEMM; the parameter keeps the Conv3d shape so reference checkpoints load."""
import torch.nn as nn

from focus_amd import ops


class PatchEmbed(nn.Module):
    def __init__(self, dim_in=3, dim_out=768, kernel=(1, 16, 16), stride=(1, 4, 4), padding=(1, 7, 7), conv_2d=False):
        super().__init__()
        if conv_2d or list(kernel) != list(stride) or any(p != 0 for p in (padding if hasattr(padding, "__iter__")
                                                                             else [padding])):
            raise NotImplementedError("hot path PatchEmbed is the non-overlapping Conv3d of Motionformer "
                                      "(video_model_builder.py:1134-1141)")
        self.kernel = tuple(kernel)
        self.proj = nn.Conv3d(dim_in, dim_out, kernel_size=kernel, stride=stride, padding=padding)
        self.compute_dtype = None     # set by the owning model (fp32 or bf16)

    def forward(self, x):
        kt, kh, kw = self.kernel
        B, Cin, T, H, W = x.shape
        cols = ops.im2col_patches(x, kt, kh, kw, self.compute_dtype or x.dtype)
        w = self.proj.weight.view(self.proj.weight.shape[0], -1)     # [D, Cin*kt*kh*kw]
        y = ops.linear(cols, w, self.proj.bias)
        return y.view(B, (T // kt) * (H // kh) * (W // kw), -1)
