"""Slot predictor: pre-LN transformer encoder over the K slots (mirror of
slowfast/models/STEVE/transformer.py:4-114).  Sequences are K<=32 tokens, so attention runs as strided
batched GEMMs + row softmax through the C ABI; the decoder half of that file is "next" (SURVEY.md 8f)."""
import torch.nn as nn

from focus_amd import ops

from .utils import linear


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model, num_heads, dropout=0.0, gain=1.0):
        super().__init__()
        assert d_model % num_heads == 0, "d_model must be divisible by num_heads"
        if dropout > 0.0:
            raise NotImplementedError("SLOTS.PREDICTOR_DROPOUT defaults to 0.0 (defaults.py:59)")
        self.d_model = d_model
        self.num_heads = num_heads
        self.proj_q = linear(d_model, d_model, bias=False)
        self.proj_k = linear(d_model, d_model, bias=False)
        self.proj_v = linear(d_model, d_model, bias=False)
        self.proj_o = linear(d_model, d_model, bias=False, gain=gain)

    def forward(self, q, k, v, attn_mask=None, residual=None):
        if attn_mask is not None:
            raise NotImplementedError("masked attention belongs to the (out of scope) decoder")
        Q = ops.linear(q, self.proj_q.weight)
        Kt = ops.linear(k, self.proj_k.weight)
        V = ops.linear(v, self.proj_v.weight)
        d = self.d_model // self.num_heads
        a = ops.small_attention(Q, Kt, V, self.num_heads, d ** -0.5)
        return ops.linear(a, self.proj_o.weight, residual=residual)


class TransformerEncoderBlock(nn.Module):
    def __init__(self, d_model, num_heads, dropout=0.0, gain=1.0, is_first=False):
        super().__init__()
        self.is_first = is_first
        self.attn_layer_norm = nn.LayerNorm(d_model)
        self.attn = MultiHeadAttention(d_model, num_heads, dropout, gain)
        self.ffn_layer_norm = nn.LayerNorm(d_model)
        self.ffn = nn.Sequential(linear(d_model, 4 * d_model, weight_init="kaiming"), nn.ReLU(),
                                 linear(4 * d_model, d_model, gain=gain), nn.Dropout(dropout))

    def forward(self, input):
        ln = self.attn_layer_norm
        if self.is_first:       # the residual stream itself is normalised (:75-78)
            input = ops.layer_norm(input, ln.weight, ln.bias, ln.eps)
            input = self.attn(input, input, input, residual=input)
        else:
            x = ops.layer_norm(input, ln.weight, ln.bias, ln.eps)
            input = self.attn(x, x, x, residual=input)
        fl = self.ffn_layer_norm
        x = ops.layer_norm(input, fl.weight, fl.bias, fl.eps)
        return ops.mlp(x, self.ffn[0].weight, self.ffn[0].bias, self.ffn[2].weight, self.ffn[2].bias,
                       residual=input, act=ops.EPI_RELU)


class TransformerEncoder(nn.Module):
    def __init__(self, num_blocks, d_model, num_heads, dropout=0.0):
        super().__init__()
        if num_blocks > 0:
            gain = (2 * num_blocks) ** (-0.5)
            self.blocks = nn.ModuleList(
                [TransformerEncoderBlock(d_model, num_heads, dropout, gain, is_first=True)] +
                [TransformerEncoderBlock(d_model, num_heads, dropout, gain, is_first=False)
                 for _ in range(num_blocks - 1)])
        else:
            self.blocks = nn.ModuleList()
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, input):
        for block in self.blocks:
            input = block(input)
        ln = self.layer_norm
        return ops.layer_norm(input, ln.weight, ln.bias, ln.eps)
