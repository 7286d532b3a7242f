"""Pre-LN transformer encoder (slot predictor) and decoder (autoregressive token decoder) of STEVE (mirror of
slowfast/models/STEVE/transformer.py:4-193).  Attention runs as strided batched GEMMs + a row softmax through the C
ABI (ops.small_attention: plain, causal, or cross-attention over the slots); Linear / LayerNorm / FFN are the fused HIP
ops of the hot path."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from focus_amd import ops

from .utils import linear


class MultiHeadAttention(nn.Module):
    """transformer.py:4-50.  attn_mask: None or the decoder's upper-triangular bool mask (applied as `causal`).
    Dropout (attention probabilities and output) is drawn with ATen in training mode."""

    def __init__(self, d_model, num_heads, dropout=0.0, gain=1.0):
        super().__init__()
        assert d_model % num_heads == 0, "d_model must be divisible by num_heads"
        self.d_model = d_model
        self.num_heads = num_heads
        self.attn_dropout = nn.Dropout(dropout)
        self.output_dropout = nn.Dropout(dropout)
        self.proj_q = linear(d_model, d_model, bias=False)
        self.proj_k = linear(d_model, d_model, bias=False)
        self.proj_v = linear(d_model, d_model, bias=False)
        self.proj_o = linear(d_model, d_model, bias=False, gain=gain)

    def forward(self, q, k, v, attn_mask=None, residual=None):
        B, T, _ = q.shape
        S = k.shape[1]
        causal = attn_mask is not None
        if causal:
            assert attn_mask.shape == (T, S) and T == S, "only the decoder's causal mask is supported"
        Q = ops.linear(q, self.proj_q.weight)
        Kt = ops.linear(k, self.proj_k.weight)
        V = ops.linear(v, self.proj_v.weight)
        d = self.d_model // self.num_heads
        p = self.attn_dropout.p
        drop = None
        if self.training and p > 0.0:
            drop = F.dropout(torch.ones(B, self.num_heads, T, S, device=q.device, dtype=q.dtype), p, True)
        a = ops.small_attention(Q, Kt, V, self.num_heads, d ** -0.5, causal=causal, drop=drop)
        if self.training and self.output_dropout.p > 0.0:
            out = self.output_dropout(ops.linear(a, self.proj_o.weight))
            return out if residual is None else residual + out
        return ops.linear(a, self.proj_o.weight, residual=residual)


def _ffn(seq, x, residual, training):
    """Linear -> ReLU -> Linear -> Dropout (+ residual)."""
    p = seq[3].p
    if training and p > 0.0:
        return residual + seq[3](ops.mlp(x, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, act=ops.EPI_RELU))
    return ops.mlp(x, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, residual=residual, act=ops.EPI_RELU)


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
            input, x = ops.layer_norm_fork(input, ln.weight, ln.bias, ln.eps)
            input = self.attn(x, x, x, residual=input)
        fl = self.ffn_layer_norm
        input, x = ops.layer_norm_fork(input, fl.weight, fl.bias, fl.eps)   # (residual-path gradient added in the LN backward)
        return _ffn(self.ffn, x, input, self.training)


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


class TransformerDecoderBlock(nn.Module):
    """transformer.py:117-166: causal self-attention, cross-attention over the slots, FFN; pre-LN (the first block
    normalises the residual stream itself)."""

    def __init__(self, max_len, d_model, num_heads, dropout=0.0, gain=1.0, is_first=False):
        super().__init__()
        self.is_first = is_first
        self.self_attn_layer_norm = nn.LayerNorm(d_model)
        self.self_attn = MultiHeadAttention(d_model, num_heads, dropout, gain)
        mask = torch.triu(torch.ones((max_len, max_len), dtype=torch.bool), diagonal=1)
        self.self_attn_mask = nn.Parameter(mask, requires_grad=False)
        self.encoder_decoder_attn_layer_norm = nn.LayerNorm(d_model)
        self.encoder_decoder_attn = MultiHeadAttention(d_model, num_heads, dropout, gain)
        self.ffn_layer_norm = nn.LayerNorm(d_model)
        self.ffn = nn.Sequential(linear(d_model, 4 * d_model, weight_init="kaiming"), nn.ReLU(),
                                 linear(4 * d_model, d_model, gain=gain), nn.Dropout(dropout))

    def forward(self, input, encoder_output):
        T = input.shape[1]
        ln = self.self_attn_layer_norm
        mask = self.self_attn_mask[:T, :T]
        if self.is_first:
            input = ops.layer_norm(input, ln.weight, ln.bias, ln.eps)
            input = self.self_attn(input, input, input, mask, residual=input)
        else:
            x = ops.layer_norm(input, ln.weight, ln.bias, ln.eps)
            input = self.self_attn(x, x, x, mask, residual=input)
        el = self.encoder_decoder_attn_layer_norm
        x = ops.layer_norm(input, el.weight, el.bias, el.eps)
        input = self.encoder_decoder_attn(x, encoder_output, encoder_output, residual=input)
        fl = self.ffn_layer_norm
        x = ops.layer_norm(input, fl.weight, fl.bias, fl.eps)
        return _ffn(self.ffn, x, input, self.training)


class TransformerDecoder(nn.Module):
    """transformer.py:169-193."""

    def __init__(self, num_blocks, max_len, d_model, num_heads, dropout=0.0):
        super().__init__()
        if num_blocks > 0:
            gain = (3 * num_blocks) ** (-0.5)
            self.blocks = nn.ModuleList(
                [TransformerDecoderBlock(max_len, d_model, num_heads, dropout, gain, is_first=True)] +
                [TransformerDecoderBlock(max_len, d_model, num_heads, dropout, gain, is_first=False)
                 for _ in range(num_blocks - 1)])
        else:
            self.blocks = nn.ModuleList()
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, input, encoder_output):
        for block in self.blocks:
            input = block(input, encoder_output)
        ln = self.layer_norm
        return ops.layer_norm(input, ln.weight, ln.bias, ln.eps)
