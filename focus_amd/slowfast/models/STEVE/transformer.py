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
        d = self.d_model // self.num_heads
        p = self.attn_dropout.p if self.training else 0.0
        bf16 = q.dtype == torch.bfloat16
        if q is k and k is v and bf16 and ((not causal and T <= 16 and self.d_model % 64 == 0) or d in (32, 48, 64)):
            # self-attention: the three projections are the column blocks of one product, which ops.small_attention's
            # one-launch kernels (the predictor over the slots) and ops.flash_attention (the decoder over the image tokens)
            # read in place
            Q, Kt, V = ops.linear_qkv(q, self.proj_q.weight, self.proj_k.weight, self.proj_v.weight)
        else:
            Q = ops.linear(q, self.proj_q.weight)
            Kt = ops.linear(k, self.proj_k.weight)
            V = ops.linear(v, self.proj_v.weight)
        if (causal or T > 16) and ops.flash_ok(Q, Kt, V, self.num_heads, causal):
            # the decoder (causal over 1024 tokens; cross-attention from them to the slots): no [T, S] probabilities and no
            # dropout mask in memory, the mask is a function of a seed drawn from torch's generator (csrc/flash_attn.hip)
            a = ops.flash_attention(Q, Kt, V, self.num_heads, d ** -0.5, causal=causal, p=p)
        else:
            drop = None
            if p > 0.0:
                drop = F.dropout(torch.ones(B, self.num_heads, T, S, device=q.device, dtype=q.dtype), p, True)
            a = ops.small_attention(Q, Kt, V, self.num_heads, d ** -0.5, causal=causal, drop=drop)
        if self.training and self.output_dropout.p > 0.0:
            return ops.dropout_add(ops.linear(a, self.proj_o.weight), residual, self.output_dropout.p, True)
        return ops.linear(a, self.proj_o.weight, residual=residual)


def _norm(ln, x):
    return ops.layer_norm(x, ln.weight, ln.bias, ln.eps)


def _make_ffn(d_model, gain, dropout):
    """The feed-forward member of both block kinds: indices 0 / 2 hold the Linear layers, 3 the dropout."""
    return nn.Sequential(linear(d_model, 4 * d_model, weight_init="kaiming"), nn.ReLU(),
                         linear(4 * d_model, d_model, gain=gain), nn.Dropout(dropout))


def _ffn(seq, x, residual, training):
    """Linear -> ReLU -> Linear -> Dropout (+ residual): one fused op, the residual in the second GEMM's epilogue."""
    up, down, drop = seq[0], seq[2], seq[3]
    if training and drop.p > 0.0:
        y = ops.mlp(x, up.weight, up.bias, down.weight, down.bias, act=ops.EPI_RELU)
        return ops.dropout_add(y, residual, drop.p, True)          # mask from a seed, rebuilt by the backward (gumbel.hip)
    return ops.mlp(x, up.weight, up.bias, down.weight, down.bias, residual=residual, act=ops.EPI_RELU)


class _BlockStack(nn.Module):
    """`blocks` (the first one flagged is_first) followed by a closing LayerNorm: the shape both transformers share.
    `branches` is the number of residual branches per block; the output projections are scaled by
    (branches * num_blocks) ** -0.5 (transformer.py:101, :175)."""

    def __init__(self, num_blocks, d_model, branches, make_block):
        super().__init__()
        gain = (branches * num_blocks) ** (-0.5) if num_blocks > 0 else 1.0
        self.blocks = nn.ModuleList([make_block(gain, i == 0) for i in range(num_blocks)])
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, input, *context):
        for block in self.blocks:
            input = block(input, *context)
        return _norm(self.layer_norm, input)


class TransformerEncoderBlock(nn.Module):
    def __init__(self, d_model, num_heads, dropout=0.0, gain=1.0, is_first=False):
        super().__init__()
        self.is_first = is_first
        self.attn_layer_norm = nn.LayerNorm(d_model)
        self.attn = MultiHeadAttention(d_model, num_heads, dropout, gain)
        self.ffn_layer_norm = nn.LayerNorm(d_model)
        self.ffn = _make_ffn(d_model, gain, dropout)

    def forward(self, input):
        ln = self.attn_layer_norm
        if self.is_first:       # the residual stream itself is normalised (:75-78)
            input = _norm(ln, input)
            input = self.attn(input, input, input, residual=input)
        else:
            input, x = ops.layer_norm_fork(input, ln.weight, ln.bias, ln.eps)
            input = self.attn(x, x, x, residual=input)
        fl = self.ffn_layer_norm
        input, x = ops.layer_norm_fork(input, fl.weight, fl.bias, fl.eps)   # (residual-path gradient added in the LN backward)
        return _ffn(self.ffn, x, input, self.training)


class TransformerEncoder(_BlockStack):
    """transformer.py:96-114 (two residual branches per block)."""

    def __init__(self, num_blocks, d_model, num_heads, dropout=0.0):
        super().__init__(num_blocks, d_model, 2,
                         lambda gain, first: TransformerEncoderBlock(d_model, num_heads, dropout, gain, is_first=first))


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
        self.ffn = _make_ffn(d_model, gain, dropout)

    def forward(self, input, encoder_output):
        T = input.shape[1]
        ln = self.self_attn_layer_norm
        mask = self.self_attn_mask[:T, :T]
        if self.is_first:
            input = _norm(ln, input)
            input = self.self_attn(input, input, input, mask, residual=input)
        else:
            x = _norm(ln, input)
            input = self.self_attn(x, x, x, mask, residual=input)
        x = _norm(self.encoder_decoder_attn_layer_norm, input)
        input = self.encoder_decoder_attn(x, encoder_output, encoder_output, residual=input)
        return _ffn(self.ffn, _norm(self.ffn_layer_norm, input), input, self.training)


class TransformerDecoder(_BlockStack):
    """transformer.py:169-193 (three residual branches per block); forward(input, encoder_output)."""

    def __init__(self, num_blocks, max_len, d_model, num_heads, dropout=0.0):
        super().__init__(num_blocks, d_model, 3, lambda gain, first: TransformerDecoderBlock(
            max_len, d_model, num_heads, dropout, gain, is_first=first))
