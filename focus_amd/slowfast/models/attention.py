"""Trajectory attention and the joint space-time block of the motion stream (mirror of slowfast/models/attention.py:353-557)
and, below them, the pooling attention of MViT (attention.py:16-352) for the MViT+ORViT variant."""
import numpy
import torch
import torch.nn as nn

from focus_amd import ops

from .common import DropPath, Mlp


class SelfAttention(nn.Module):
    """Joint MHSA over a short token sequence (attention.py:355-385)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, thw, residual=None):
        C = x.shape[-1]
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        a = ops.small_attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.num_heads, self.scale)
        return ops.linear(a, self.proj.weight, self.proj.bias, residual=residual), thw


class SeltAttentionBlock(nn.Module):
    """Pre-LN block used by the ORViT motion stream (attention.py:388-432)."""

    def __init__(self, dim, num_heads=None, mlp_ratio=4.0, qkv_bias=False, drop_rate=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, has_cls_embed=True):
        super().__init__()
        self.dim = dim
        self.norm1 = norm_layer(dim)
        self.attn = SelfAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_drop=drop_rate)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.has_cls_embed = has_cls_embed
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), out_features=dim, act_layer=act_layer,
                       drop=drop_rate)

    def forward(self, x, metadata, thw_shape):
        n1, n2 = self.norm1, self.norm2
        if isinstance(self.drop_path, nn.Identity) or not self.training:
            x, thw = self.attn(ops.layer_norm(x, n1.weight, n1.bias, n1.eps), thw_shape, residual=x)
            x = self.mlp(ops.layer_norm(x, n2.weight, n2.bias, n2.eps), residual=x)
            return x, thw
        y, thw = self.attn(ops.layer_norm(x, n1.weight, n1.bias, n1.eps), thw_shape)
        x = x + self.drop_path(y)
        x = x + self.drop_path(self.mlp(ops.layer_norm(x, n2.weight, n2.bias, n2.eps)))
        return x, thw


class TrajectoryAttention(nn.Module):
    """Motionformer trajectory attention (attention.py:479-557).

    space step : per (query, frame) softmax over that frame's P keys -> x~ [B,S,F,C]  (fused HIP)
    time step  : q2 = proj_q(diag(x~)), k2 = proj_kv(x~)[:C]; softmax over F; out = sum_f a2 * x~
                 (use_original_code=True, the only mode the reference ever runs; the v2 half of proj_kv
                  is dead there -- zero output use, zero gradient -- and is not computed)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, use_original_code=True):
        super().__init__()
        if attn_drop > 0.0 or proj_drop > 0.0:
            raise NotImplementedError("MF.ATTN_DROPOUT / MF.DROP > 0 is not used by any hot-path config")
        if not use_original_code:
            raise NotImplementedError("use_original_code=False is never instantiated by the reference "
                                      "(attention.py:449,480; orvit.py:83-89)")
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj_q = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj_kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_original_code = use_original_code

    def forward(self, x, thw_prev, with_cls_token=True, residual=None):
        if not with_cls_token:
            raise NotImplementedError("with_cls_token=False is not reachable from Motionformer/ORViT")
        B, N, C = x.shape
        F_, P = thw_prev[0], thw_prev[1] * thw_prev[2]
        h = self.num_heads
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)                       # :506
        xt, xdiag, cls_out = ops.traj_space(qkv, F_, P, h)                         # :509-535
        q2 = ops.linear(xdiag, self.proj_q.weight, self.proj_q.bias)               # :536 (scale applied in-kernel)
        # temporal step (:537-549) as one autograd node whose output already is cat(cls_out, out) (the attention rows
        # are written into tokens 1.. in place).  bf16 / head dim 64: k2 = proj_kv(x~)[:C] is never formed (the logits
        # are re-associated as (Wk[h]^T q2) . x~, csrc/traj_time2.hip); otherwise k2 GEMM + time kernels.
        if ops.traj_time2_ok(xt, h):
            y_in = ops.traj_time2_block(q2, xt, self.proj_kv.weight, self.proj_kv.bias, cls_out, h)
        else:
            y_in = ops.traj_time_block(q2, xt, self.proj_kv.weight, self.proj_kv.bias, cls_out, h)
        return ops.linear(y_in, self.proj.weight, self.proj.bias, residual=residual), thw_prev


class TrajectoryAttentionBlock(nn.Module):
    """attention.py:443-476."""

    def __init__(self, cfg=None, dim=768, num_heads=12, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_original_code=True):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = TrajectoryAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop,
                                        proj_drop=drop, use_original_code=use_original_code)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _linear_params(self):
        a, m = self.attn, self.mlp
        return [a.qkv.weight, a.qkv.bias, a.proj_q.weight, a.proj_q.bias, a.proj.weight, a.proj.bias,
                m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]

    def forward(self, x, metadata, thw, with_cls_token=True):
        # the five Linear weight gradients of the block are formed in one launch (ops.wgrad_group)
        with ops.wgrad_group(self._linear_params()):
            return self._forward(x, metadata, thw, with_cls_token)

    def _forward(self, x, metadata, thw, with_cls_token=True):
        n1, n2 = self.norm1, self.norm2
        if isinstance(self.drop_path, nn.Identity) or not self.training:
            # residual adds ride in the proj / fc2 GEMM epilogues
            xr, h = ops.layer_norm_fork(x, n1.weight, n1.bias, n1.eps)
            x = self.attn(h, thw, with_cls_token, residual=xr)[0]
            xr, h = ops.layer_norm_fork(x, n2.weight, n2.bias, n2.eps)
            x = self.mlp(h, residual=xr)
            return x, thw
        # training with stochastic depth: x + mask_b/keep * branch in one fused pass per residual
        # (layer_norm_fork: the residual-path gradient is added inside the LayerNorm backward kernel)
        dp = self.drop_path.drop_prob
        xr, h = ops.layer_norm_fork(x, n1.weight, n1.bias, n1.eps)
        x = ops.residual_drop_path(xr, self.attn(h, thw, with_cls_token)[0], dp, True)
        xr, h = ops.layer_norm_fork(x, n2.weight, n2.bias, n2.eps)
        x = ops.residual_drop_path(xr, self.mlp(h), dp, True)
        return x, thw


# --------------------------------------------------------------------------------------------------
# Multiscale (pooling) attention: attention.py:16-352.  Linear / LayerNorm / Mlp / the attention products run on the HIP
# kernels (ops.linear, ops.layer_norm, ops.mlp, ops.small_attention or ops.flash_attention); the pooling operators
# themselves -- depth-wise Conv3d, MaxPool3d, AvgPool3d -- stay with ATen / MIOpen in fp32, like the dVAE convolutions.
# --------------------------------------------------------------------------------------------------
def attention_pool(tensor, pool, thw_shape, has_cls_embed=True, norm=None):
    """attention.py:16-50: [B, heads, L, C] (or [B, L, C]) -> pooled over the (T, H, W) grid, cls token passed around."""
    if pool is None:
        return tensor, thw_shape
    tensor_dim = tensor.ndim
    if tensor_dim == 3:
        tensor = tensor.unsqueeze(1)
    elif tensor_dim != 4:
        raise NotImplementedError("Unsupported input dimension %s" % (tuple(tensor.shape),))
    if has_cls_embed:
        cls_tok, tensor = tensor[:, :, :1, :], tensor[:, :, 1:, :]
    B, N, L, C = tensor.shape
    T, H, W = thw_shape
    dt = tensor.dtype
    tensor = tensor.reshape(B * N, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    tensor = pool(tensor.float()).to(dt)
    thw_shape = [tensor.shape[2], tensor.shape[3], tensor.shape[4]]
    L_pooled = tensor.shape[2] * tensor.shape[3] * tensor.shape[4]
    tensor = tensor.reshape(B, N, C, L_pooled).transpose(2, 3)
    if has_cls_embed:
        tensor = torch.cat((cls_tok, tensor), dim=2)
    if norm is not None:
        tensor = ops.layer_norm(tensor.contiguous(), norm.weight, norm.bias, norm.eps)
    if tensor_dim == 3:
        tensor = tensor.squeeze(1)
    return tensor, thw_shape


class MultiScaleAttention(nn.Module):
    """attention.py:52-258 (the (x, q_shape) result; the val_output_type probes of :247-256 are evaluation tooling)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, drop_rate=0.0, kernel_q=(1, 1, 1), kernel_kv=(1, 1, 1),
                 stride_q=(1, 1, 1), stride_kv=(1, 1, 1), norm_layer=nn.LayerNorm, has_cls_embed=True, mode="conv",
                 pool_first=False, ignore_111_kv_kernel=False):
        super().__init__()
        self.pool_first = pool_first
        self.drop_rate = drop_rate
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.has_cls_embed = has_cls_embed
        padding_q = [int(q // 2) for q in kernel_q]
        padding_kv = [int(kv // 2) for kv in kernel_kv]
        self.pool_kv = len(kernel_kv) > 0
        if ignore_111_kv_kernel:
            self.pool_kv = self.pool_kv and (tuple(stride_kv) != (1, 1, 1))
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.k = nn.Linear(dim, dim, bias=qkv_bias)
        self.v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if drop_rate > 0.0:
            self.proj_drop = nn.Dropout(drop_rate)
        if numpy.prod(kernel_q) == 1 and numpy.prod(stride_q) == 1:        # (1,1,1) pooling is no pooling
            kernel_q = ()
        if numpy.prod(kernel_kv) == 1 and numpy.prod(stride_kv) == 1:
            kernel_kv = ()
        if mode in ("avg", "max"):
            pool_op = nn.MaxPool3d if mode == "max" else nn.AvgPool3d
            self.pool_q = pool_op(kernel_q, stride_q, padding_q, ceil_mode=False) if len(kernel_q) > 0 else None
            self.pool_k = pool_op(kernel_kv, stride_kv, padding_kv, ceil_mode=False) if len(kernel_kv) > 0 else None
            self.pool_v = pool_op(kernel_kv, stride_kv, padding_kv, ceil_mode=False) if len(kernel_kv) > 0 else None
        elif mode == "conv":
            def conv(kernel, stride, padding):
                return nn.Conv3d(head_dim, head_dim, kernel, stride=stride, padding=padding, groups=head_dim, bias=False)
            self.pool_q = conv(kernel_q, stride_q, padding_q) if len(kernel_q) > 0 else None
            self.norm_q = norm_layer(head_dim) if len(kernel_q) > 0 else None
            self.pool_k = conv(kernel_kv, stride_kv, padding_kv) if self.pool_kv else None
            self.norm_k = norm_layer(head_dim) if self.pool_kv else None
            self.pool_v = conv(kernel_kv, stride_kv, padding_kv) if self.pool_kv else None
            self.norm_v = norm_layer(head_dim) if self.pool_kv else None
        else:
            raise NotImplementedError("Unsupported model %s" % mode)

    def _heads(self, t, B, n):
        return t.reshape(B, n, self.num_heads, -1).permute(0, 2, 1, 3)

    def forward(self, x, thw_shape):
        B, N, C = x.shape
        if self.pool_first:
            q = k = v = self._heads(x, B, N)
        else:
            q = self._heads(ops.linear(x, self.q.weight, self.q.bias), B, N)
            k = self._heads(ops.linear(x, self.k.weight, self.k.bias), B, N)
            v = self._heads(ops.linear(x, self.v.weight, self.v.bias), B, N)
        q, q_shape = attention_pool(q, self.pool_q, thw_shape, self.has_cls_embed, getattr(self, "norm_q", None))
        k, k_shape = attention_pool(k, self.pool_k, thw_shape, self.has_cls_embed, getattr(self, "norm_k", None))
        v, v_shape = attention_pool(v, self.pool_v, thw_shape, self.has_cls_embed, getattr(self, "norm_v", None))
        # back to [B, n, C] rows: the layout the attention kernels read (heads = column blocks)
        q = q.permute(0, 2, 1, 3).reshape(B, -1, C)
        k = k.permute(0, 2, 1, 3).reshape(B, -1, C)
        v = v.permute(0, 2, 1, 3).reshape(B, -1, C)
        if self.pool_first:
            q = ops.linear(q, self.q.weight, self.q.bias)
            k = ops.linear(k, self.k.weight, self.k.bias)
            v = ops.linear(v, self.v.weight, self.v.bias)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        if ops.flash_ok(q, k, v, self.num_heads, False):
            a = ops.flash_attention(q, k, v, self.num_heads, self.scale)
        else:
            a = ops.small_attention(q, k, v, self.num_heads, self.scale)
        x = ops.linear(a, self.proj.weight, self.proj.bias)
        if self.drop_rate > 0.0:
            x = self.proj_drop(x)
        return x, q_shape


class MultiScaleBlock(nn.Module):
    """attention.py:260-352."""

    def __init__(self, dim, dim_out, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, up_rate=None, kernel_q=(1, 1, 1), kernel_kv=(1, 1, 1),
                 stride_q=(1, 1, 1), stride_kv=(1, 1, 1), mode="conv", has_cls_embed=True, pool_first=False,
                 ignore_111_kv_kernel=False, cfg=None):
        super().__init__()
        self.dim = dim
        self.dim_out = dim_out
        self.norm1 = norm_layer(dim)
        kernel_skip = [s + 1 if s > 1 else s for s in stride_q]
        stride_skip = stride_q
        padding_skip = [int(skip // 2) for skip in kernel_skip]
        self.attn = MultiScaleAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, drop_rate=drop_rate, kernel_q=kernel_q,
                                        kernel_kv=kernel_kv, stride_q=stride_q, stride_kv=stride_kv, norm_layer=nn.LayerNorm,
                                        has_cls_embed=has_cls_embed, mode=mode, pool_first=pool_first,
                                        ignore_111_kv_kernel=ignore_111_kv_kernel)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.has_cls_embed = has_cls_embed
        mlp_dim_out = dim * up_rate if up_rate is not None and up_rate > 1 else dim_out
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, out_features=mlp_dim_out, act_layer=act_layer,
                       drop=drop_rate)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)
        self.pool_skip = nn.MaxPool3d(kernel_skip, stride_skip, padding_skip, ceil_mode=False) if len(kernel_skip) > 0 else None

    def forward(self, x, metadata, thw_shape):
        n1, n2 = self.norm1, self.norm2
        x_block, thw_shape_new = self.attn(ops.layer_norm(x, n1.weight, n1.bias, n1.eps), thw_shape)
        x_res, _ = attention_pool(x, self.pool_skip, thw_shape, has_cls_embed=self.has_cls_embed)
        x = x_res + self.drop_path(x_block)
        x_norm = ops.layer_norm(x, n2.weight, n2.bias, n2.eps)
        if self.dim != self.dim_out:
            x = ops.linear(x_norm, self.proj.weight, self.proj.bias)
        if isinstance(self.drop_path, nn.Identity) or not self.training:
            return self.mlp(x_norm, residual=x.contiguous()), thw_shape_new
        return x + self.drop_path(self.mlp(x_norm)), thw_shape_new
