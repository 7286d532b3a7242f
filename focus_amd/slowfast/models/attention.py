"""Trajectory attention and the joint space-time block of the motion stream
(mirror of slowfast/models/attention.py:353-557; the MViT part of that file is out of scope)."""
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
