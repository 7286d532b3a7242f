"""ORViT object-region block (mirror of slowfast/models/ORViT/orvit.py:39-269)."""
import torch
from torch import nn

from focus_amd import ops
from focus_amd.slowfast.models.attention import SeltAttentionBlock, TrajectoryAttention

from ..common import DropPath
from .utils import Mlp, ObjectsCrops, box2spatial_layout


import os

# The motion stream (orvit.py:159-161) depends on the box coordinates only: 32 tokens per clip, i.e. GEMMs of 256 rows
# that occupy a handful of CUs for 20-60 us each (1.3 ms per bench step when serialised with the main branch).  It is
# issued on a second HIP stream, forked before the RoI / attention work of the block and joined before motion_mlp, so
# those launches run beside the block's large kernels; autograd replays the same fork / join for its backward
# (33.3 vs 34.3 ms per bench step).  Under DistributedDataParallel the reducer orders a bucket's all-reduce after the
# stream of the LAST gradient hook only, which does not cover gradients produced on another stream: both streams are
# registered with focus_amd.parallel, whose communication hook joins them before every bucket's collective.
_SIDE_STREAMS = {}
_USE_SIDE_STREAM = os.environ.get("FOCUS_MOTION_SIDE_STREAM", "1") != "0"
_COMMUTE_PATCH_TO_D = os.environ.get("FOCUS_ORVIT_COMMUTE", "1") != "0"      # patch_to_d[0] before RoIAlign (see forward)


def _side_stream(device):
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def _relu_pair(seq, x, dtype=None):
    """nn.Sequential(Linear(no bias), ReLU, Linear(no bias), ReLU) on the 4-d box coordinates: the first layer
    (K=4) stays in fp32 on the raw coordinates, the second runs in the compute dtype (MFMA when bf16)."""
    hid = torch.relu(ops.linear(x.float(), seq[0].weight))
    if dtype is not None and dtype != hid.dtype:
        hid = hid.to(dtype)
    return torch.relu(ops.linear(hid, seq[2].weight))


class ORViT(nn.Module):
    def __init__(self, cfg, dim=768, dim_out=None, num_heads=12, attn_type="trajectory", mlp_ratio=4.0,
                 qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 use_original_code=False, nb_frames=None):
        super().__init__()
        self.cfg = cfg
        self.in_dim = dim
        self.dim = dim
        self.nb_frames = nb_frames
        self.with_cls_token = True
        self.with_motion_stream = cfg.ORVIT.USE_MOTION_STREAM

        self.crop_layer = ObjectsCrops(cfg)
        self.patch_to_d = nn.Sequential(nn.Linear(dim, dim // 2, bias=False), nn.ReLU(inplace=True),
                                        nn.Linear(dim // 2, dim, bias=False), nn.ReLU())
        self.box_categories = nn.Parameter(torch.zeros(nb_frames, cfg.ORVIT.O, dim))
        self.c_coord_to_feature = nn.Sequential(nn.Linear(4, dim // 2, bias=False), nn.ReLU(inplace=True),
                                                nn.Linear(dim // 2, dim, bias=False), nn.ReLU())
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm1 = norm_layer(dim)
        self.norm2 = norm_layer(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.attn = TrajectoryAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop,
                                        proj_drop=drop)
        if self.with_motion_stream:
            self.motion_stream = MotionStream(cfg, dim=dim, num_heads=num_heads,
                                              attn_type=cfg.ORVIT.MOTION_STREAM_ATTN_TYPE, mlp_ratio=mlp_ratio,
                                              qkv_bias=qkv_bias, drop=drop, attn_drop=attn_drop, drop_path=drop_path,
                                              act_layer=act_layer, norm_layer=norm_layer, nb_frames=nb_frames)
            self.motion_mlp = Mlp(in_features=cfg.ORVIT.MOTION_STREAM_DIM if cfg.ORVIT.MOTION_STREAM_DIM > 0 else dim,
                                  hidden_features=mlp_hidden_dim, out_features=dim, act_layer=act_layer, drop=drop)
        if cfg.ORVIT.INIT_WEIGHTS:
            self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        else:
            for p in m.parameters():
                nn.init.normal_(p, std=0.02)

    def _linear_params(self):
        a, m, mm, p2d = self.attn, self.mlp, getattr(self, "motion_mlp", None), self.patch_to_d
        ps = [a.qkv.weight, a.qkv.bias, a.proj_q.weight, a.proj_q.bias, a.proj.weight, a.proj.bias,
              m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, p2d[0].weight, p2d[2].weight]
        if mm is not None:
            ps += [mm.fc1.weight, mm.fc1.bias, mm.fc2.weight, mm.fc2.bias]
        return ps

    def forward(self, x, metadata, thw):
        # the Linear weight gradients of the block (attention, MLPs, patch_to_d) are formed in grouped launches
        with ops.wgrad_group(self._linear_params()):
            return self._forward(x, metadata, thw)

    def _forward(self, x, metadata, thw):
        box_tensors = metadata["orvit_bboxes"]
        assert box_tensors is not None
        BS, _, d = x.shape
        T, H, W = thw
        assert T == self.nb_frames
        Tratio = box_tensors.shape[1] // T
        box_tensors = box_tensors[:, ::Tratio].float()                     # [BS,T,O,4]
        O = box_tensors.shape[-2]
        HW = H * W

        motion_emb, side = None, None
        if self.with_motion_stream:
            if x.is_cuda and _USE_SIDE_STREAM:
                main, side = torch.cuda.current_stream(), _side_stream(x.device)
                if torch.distributed.is_available() and torch.distributed.is_initialized():
                    from focus_amd import parallel
                    parallel.note_grad_stream(main)
                    parallel.note_grad_stream(side)
                side.wait_stream(main)
                box_tensors.record_stream(side)
                with torch.cuda.stream(side):
                    motion_emb = self.motion_stream(box_tensors, H, W, dtype=x.dtype)        # [BS,T*H*W,d]
            else:
                motion_emb = self.motion_stream(box_tensors, H, W, dtype=x.dtype)

        # object tokens: RoIAlign -> patch_to_d -> max over the RoI cells (:135-139).  patch_to_d's first Linear has no bias,
        # so it commutes with the bilinear sampling: it is applied ONCE to the T*HW patch tokens of the stream
        # (O = 4 times fewer rows than the RoI cells), the crops are sampled in dim/2 channels (half the gather bytes, half
        # the crop tensor, half the RoIAlign backward) with the ReLU fused into the sampling kernels; only the second
        # Linear runs on the RoI cells.  Same values up to rounding order.
        p2d = self.patch_to_d
        if _COMMUTE_PATCH_TO_D:
            z = ops.linear(x, p2d[0].weight)                                                 # [BS, 1+T*HW, d/2]
            crops = self.crop_layer.crop_stream(z, box_tensors, T, H, W, relu=True)         # [BS*T*O, HW, d/2]
            pre = ops.linear(crops, p2d[2].weight)                                           # last ReLU commutes with max
        else:                                                                                # the reference's order (:135-137)
            crops = self.crop_layer.crop_stream(x, box_tensors, T, H, W)                    # [BS*T*O, HW, d]
            pre = ops.mlp(crops, p2d[0].weight, None, p2d[2].weight, None, act=ops.EPI_RELU)
        obj = torch.relu(ops.cell_amax(pre)).view(BS, T, O, d)
        box_emb = _relu_pair(self.c_coord_to_feature, box_tensors, x.dtype)
        obj = obj + self.box_categories.to(x.dtype) + box_emb                               # :141-143

        # :145-147 (two cats) as one pass; :152-169 (slice, reshape copy, motion residual, cat, drop-path, add) as another
        all_tokens = ops.orvit_assemble(x, obj, T, HW)                                      # [BS, 1+T*(HW+O), d]
        n1 = self.norm1
        y, _ = self.attn(ops.layer_norm(all_tokens, n1.weight, n1.bias, n1.eps), [T, HW + O, 1])
        mm = None
        if self.with_motion_stream:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                motion_emb.record_stream(torch.cuda.current_stream())
            mm = self.motion_mlp(motion_emb)                                                 # :162-163
        dp = 0.0 if isinstance(self.drop_path, nn.Identity) else self.drop_path.drop_prob
        x = ops.orvit_merge(x, y, mm, T, HW, dp, self.training)                              # :169
        n2 = self.norm2
        xr, h = ops.layer_norm_fork(x, n2.weight, n2.bias, n2.eps)
        if dp == 0.0 or not self.training:
            x = self.mlp(h, residual=xr)                                                     # :170, add in the fc2 epilogue
        else:
            x = ops.residual_drop_path(xr, self.mlp(h), dp, True)
        return x, thw


class Object2Spatial(nn.Module):
    def __init__(self, cfg, _type):
        super().__init__()
        self.cfg = cfg
        self._type = _type

    def forward(self, all_features, context, boxes, H, W, t_avg_pooling=False):
        BS, T, O, d = all_features.shape
        if self._type != "layout":
            raise NotImplementedError("%s: only the 'layout' mapping is used (orvit.py:252)" % self._type)
        ret = box2spatial_layout(boxes, all_features, H, W).permute(0, 2, 3, 4, 1)    # [BS,T,H,W,d]
        if t_avg_pooling:
            Tratio = int(T / self.cfg.MF.TEMPORAL_RESOLUTION)
            if Tratio > 1:
                ret = ret.reshape(BS, -1, Tratio, H, W, d).mean(2)
        return ret.flatten(1, 3)


class MotionStream(nn.Module):
    def __init__(self, cfg, dim=768, num_heads=12, attn_type="trajectory", mlp_ratio=4.0, qkv_bias=False, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, nb_frames=None):
        super().__init__()
        self.cfg = cfg
        self.in_dim = cfg.ORVIT.MOTION_STREAM_DIM if cfg.ORVIT.MOTION_STREAM_DIM > 0 else dim
        self.dim = dim
        self.nb_frames = nb_frames
        if cfg.ORVIT.MOTION_STREAM_SEP_POS_EMB:
            self.box_categories_T = nn.Parameter(torch.zeros(nb_frames, 1, self.in_dim))
            self.box_categories_O = nn.Parameter(torch.zeros(1, cfg.ORVIT.O, self.in_dim))
        else:
            self.box_categories = nn.Parameter(torch.zeros(nb_frames, cfg.ORVIT.O, self.in_dim))
        self.c_coord_to_feature = nn.Sequential(nn.Linear(4, self.in_dim // 2, bias=False), nn.ReLU(inplace=True),
                                                nn.Linear(self.in_dim // 2, self.in_dim, bias=False), nn.ReLU())
        self.attn_type = attn_type
        if attn_type != "joint":
            raise NotImplementedError("ORVIT.MOTION_STREAM_ATTN_TYPE must be 'joint' (orvit.py:238)")
        self.attn = SeltAttentionBlock(dim=self.in_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                       drop_rate=attn_drop, drop_path=drop_path, act_layer=act_layer,
                                       norm_layer=norm_layer)
        self.obj2spatial = Object2Spatial(cfg, _type="layout")

    def forward(self, box_tensors, H, W, dtype=None):
        BS = box_tensors.shape[0]
        dtype = dtype or box_tensors.dtype
        box_emb = _relu_pair(self.c_coord_to_feature, box_tensors, dtype)
        if self.cfg.ORVIT.MOTION_STREAM_SEP_POS_EMB:
            shape = (self.nb_frames, self.cfg.ORVIT.O, self.in_dim)
            cat = self.box_categories_T.expand(shape) + self.box_categories_O.expand(shape)
        else:
            cat = self.box_categories
        box_emb = cat.to(dtype).unsqueeze(0) + box_emb                      # [BS,T,O,d]
        oshape = box_emb.shape
        box_emb, _ = self.attn(box_emb.flatten(1, -2), None, None)
        box_emb = box_emb.reshape(oshape)
        return self.obj2spatial(box_emb, None, box_tensors, H, W, t_avg_pooling=True)   # [BS,T*H*W,d]
