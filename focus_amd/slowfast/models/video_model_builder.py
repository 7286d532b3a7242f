"""Motionformer backbone with ORViT blocks (mirror of slowfast/models/video_model_builder.py:1103-1353) and, below it, the
MViT backbone with ORViT blocks (:765-1101).  The CNN builders of that file (SlowFast / ResNet / X3D) are out of scope."""
import math
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_

from focus_amd import ops

from . import stem_helper
from .attention import MultiScaleBlock, TrajectoryAttentionBlock
from .build import MODEL_REGISTRY
from .ORViT import ORViT


@MODEL_REGISTRY.register()
class Motionformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.img_size = cfg.DATA.TRAIN_CROP_SIZE
        self.patch_size = cfg.MF.PATCH_SIZE
        self.in_chans = cfg.MF.CHANNELS
        self.num_classes = [97, 300] if cfg.TRAIN.DATASET == "epickitchens" else cfg.MODEL.NUM_CLASSES
        self.embed_dim = cfg.MF.EMBED_DIM
        self.depth = cfg.MF.DEPTH
        self.num_heads = cfg.MF.NUM_HEADS
        self.mlp_ratio = cfg.MF.MLP_RATIO
        self.qkv_bias = cfg.MF.QKV_BIAS
        self.drop_rate = cfg.MF.DROP
        self.drop_path_rate = cfg.MF.DROP_PATH
        self.head_dropout = cfg.MF.HEAD_DROPOUT
        self.video_input = cfg.MF.VIDEO_INPUT
        self.temporal_resolution = cfg.MF.TEMPORAL_RESOLUTION
        self.use_mlp = cfg.MF.USE_MLP
        self.num_features = self.embed_dim
        self.attn_drop_rate = cfg.MF.ATTN_DROPOUT
        self.head_act = cfg.MF.HEAD_ACT
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        if not self.video_input or cfg.MF.POS_EMBED != "separate":
            raise NotImplementedError("hot path = video input with separate space/time position embeddings")
        # bf16 storage + fp32 accumulation when TRAIN.MIXED_PRECISION (the reference autocasts to fp16)
        self.compute_dtype = torch.bfloat16 if cfg.TRAIN.MIXED_PRECISION else torch.float32
        # build-owned key (BASELINE configs[4]): the Linear weights of the blocks are multiplied as OCP e4m3 copies with one
        # scale per tensor, activations stay bf16 (focus_amd.ops.fp8_weights); needs the bf16 compute dtype
        self.fp8_weights = bool(cfg.TRAIN.get("FP8_WEIGHTS", False)) and cfg.TRAIN.MIXED_PRECISION

        k = [cfg.MF.PATCH_SIZE_TEMP, self.patch_size, self.patch_size]
        self.patch_embed_3d = stem_helper.PatchEmbed(dim_in=self.in_chans, dim_out=self.embed_dim, kernel=k, stride=k,
                                                     padding=0, conv_2d=False)
        self.patch_embed_3d.compute_dtype = self.compute_dtype
        self.patch_embed_3d.num_patches = (224 // self.patch_size) ** 2
        self.patch_embed_3d.proj.weight.data = torch.zeros_like(self.patch_embed_3d.proj.weight.data)
        self.num_patches = self.patch_embed_3d.num_patches * self.temporal_resolution

        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        trunc_normal_(self.cls_token, std=0.02)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed_3d.num_patches + 1, self.embed_dim))
        self.pos_drop = nn.Dropout(p=cfg.MF.POS_DROPOUT)
        trunc_normal_(self.pos_embed, std=0.02)
        self.temp_embed = nn.Parameter(torch.zeros(1, self.temporal_resolution, self.embed_dim))

        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        blocks = []
        for i in range(self.depth):
            if i in cfg.ORVIT.LAYERS:
                blocks.append(ORViT(cfg=cfg, dim=self.embed_dim, num_heads=self.num_heads, mlp_ratio=self.mlp_ratio,
                                    qkv_bias=self.qkv_bias, drop=self.drop_rate, attn_drop=self.attn_drop_rate,
                                    norm_layer=norm_layer, nb_frames=self.temporal_resolution))
            else:
                blocks.append(TrajectoryAttentionBlock(cfg=cfg, dim=self.embed_dim, num_heads=self.num_heads,
                                                       mlp_ratio=self.mlp_ratio, qkv_bias=self.qkv_bias,
                                                       drop=self.drop_rate, attn_drop=self.attn_drop_rate,
                                                       drop_path=dpr[i], norm_layer=norm_layer))
        self.blocks = nn.ModuleList(blocks)
        self.norm = norm_layer(self.embed_dim)

        if self.use_mlp:
            act = {"tanh": nn.Tanh, "gelu": nn.GELU}.get(self.head_act, nn.ReLU)()
            self.pre_logits = nn.Sequential(OrderedDict([("fc", nn.Linear(self.embed_dim, self.embed_dim)),
                                                         ("act", act)]))
        else:
            self.pre_logits = nn.Identity()
        self.head_drop = nn.Dropout(p=self.head_dropout)
        if isinstance(self.num_classes, list) and len(self.num_classes) > 1:
            for a, n in enumerate(self.num_classes):
                setattr(self, "head%d" % a, nn.Linear(self.embed_dim, n))
        else:
            self.head = nn.Linear(self.embed_dim, self.num_classes) if self.num_classes > 0 else nn.Identity()

        self._bicubic = {}            # (grid, patches, device) -> the bicubic resampling matrix of _spatial_pos
        self.init_weights()
        self.apply(self._init_weights)

    def init_weights(self):
        for _, p in self.named_parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "temp_embed"}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=""):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def _spatial_pos(self, npatch):
        """pos_embed[:,1:] resampled bicubically when the crop is not 224 (:1285-1300)."""
        pos = self.pos_embed
        n0 = pos.shape[1] - 1
        if self.cfg.DATA.TRAIN_CROP_SIZE == 224:
            return pos[0]
        g = int(math.sqrt(n0))
        # Bicubic resampling is a fixed linear map of the g x g grid: its [npatch, n0] matrix is read off ATen's own
        # interpolate applied to the identity (once per size and device: the same taps, hence the same values up to the order
        # of a 16-term fp32 sum) and applied as one small product.  ATen's upsample_bicubic2d / its backward on the
        # [1, 768, 14, 14] tensor took 2.5 + 6.6 ms of EVERY 48 ms HR step (profiles/r03_kernel_summary_hr.txt).
        key = (g, npatch, pos.device)
        m = self._bicubic.get(key)
        if m is None:
            eye = torch.eye(n0, device=pos.device, dtype=torch.float32).reshape(1, n0, g, g)
            with torch.no_grad():
                up = torch.nn.functional.interpolate(eye, scale_factor=math.sqrt(npatch / n0), mode="bicubic")
            m = self._bicubic[key] = up.reshape(n0, -1).t().contiguous()          # [npatch, n0]
        sp = m @ pos[0, 1:].float()
        return torch.cat([pos[0, :1], sp.to(pos.dtype)], dim=0)

    def forward_features(self, x, metadata):
        x = x[0]
        tok = self.patch_embed_3d(x)                                         # [B, T*H*W, D]  K1
        npatch = tok.shape[1] // self.temporal_resolution
        x = ops.embed_assemble(tok, self.cls_token.view(-1), self._spatial_pos(npatch), self.temp_embed[0])  # K2
        if self.pos_drop.p > 0:
            x = self.pos_drop(x)
        side = int(npatch ** 0.5)
        thw = [self.temporal_resolution, side, side]
        with ops.fp8_weights(self.fp8_weights):
            for blk in self.blocks:
                x, _ = blk(x, metadata, thw)
        n = self.norm
        x = ops.layer_norm(x[:, 0].contiguous(), n.weight, n.bias, n.eps)    # LN is per token: only cls is needed
        if self.use_mlp:
            x = self.pre_logits.act(ops.linear(x, self.pre_logits.fc.weight, self.pre_logits.fc.bias))
        return x

    def forward(self, x, metadata):
        x = self.forward_features(x, metadata)
        x = self.head_drop(x)
        if isinstance(self.num_classes, list) and len(self.num_classes) > 1:
            output = []
            for hd in range(len(self.num_classes)):
                m = getattr(self, "head%d" % hd)
                o = ops.linear(x, m.weight, m.bias).float()
                output.append(o if self.training else torch.softmax(o, dim=-1))
            return output[0], {"verb": output[0], "noun": output[1]}
        x = ops.linear(x, self.head.weight, self.head.bias).float()
        return x if self.training else torch.softmax(x, dim=-1)


def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:31-44: channel / head counts after a stage multiplier, rounded to a multiple of `divisor`
    (never below 90 % of the exact product)."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


class TransformerBasicHead(nn.Module):
    """head_helper.py:363-419: (dropout,) Linear, and the activation outside training.  One head or a dict of heads."""

    def __init__(self, dim_in, num_classes, dropout_rate=0.0, act_func="softmax"):
        super().__init__()
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        if isinstance(num_classes, dict):
            self.projection = nn.ModuleDict({k: nn.Linear(dim_in, num_classes[k], bias=True) for k in num_classes})
        else:
            self.projection = nn.Linear(dim_in, num_classes, bias=True)
        if act_func == "softmax":
            self.act = nn.Softmax(dim=1)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        else:
            raise NotImplementedError("{} is not supported as an activation function.".format(act_func))

    def forward(self, x):
        if hasattr(self, "dropout"):
            x = self.dropout(x)
        if isinstance(self.projection, nn.ModuleDict):
            extra_preds = {k: ops.linear(x, m.weight, m.bias).float() for k, m in self.projection.items()}
            if not self.training:
                extra_preds = {k: self.act(v) for k, v in extra_preds.items()}
            return torch.zeros(1).to(x.device), extra_preds
        x = ops.linear(x, self.projection.weight, self.projection.bias).float()
        return x if self.training else self.act(x)


@MODEL_REGISTRY.register()
class MViT(nn.Module):
    """Multiscale Vision Transformer with ORViT blocks (video_model_builder.py:765-1101): an ORViT block takes the place of
    the MultiScaleBlock at the layers in ORVIT.LAYERS (:915-928) or runs beside it at the layers in ORVIT.ADD_LAYERS, its
    output added to the block's (:952-972, :1077-1082).  The detection head (:975-986, RoI head over AVA boxes) is not part
    of this path."""

    def __init__(self, cfg):
        super().__init__()
        assert cfg.DATA.TRAIN_CROP_SIZE == cfg.DATA.TEST_CROP_SIZE
        self.cfg = cfg
        mv = cfg.MVIT
        if cfg.DETECTION.ENABLE:
            raise NotImplementedError("DETECTION.ENABLE (ResNetRoIHead) is outside the hot path")
        pool_first = mv.POOL_FIRST
        spatial_size, temporal_size = cfg.DATA.TRAIN_CROP_SIZE, cfg.DATA.NUM_FRAMES
        in_chans = cfg.DATA.INPUT_CHANNEL_NUM[0]
        if mv.PATCH_2D:
            raise NotImplementedError("MVIT.PATCH_2D (image models) is not part of the video path")
        self.patch_stride = list(mv.PATCH_STRIDE)
        num_classes = {"verb": 97, "noun": 300} if cfg.TRAIN.DATASET == "epickitchens" else cfg.MODEL.NUM_CLASSES  # misc.get_num_classes
        embed_dim, num_heads, depth = mv.EMBED_DIM, mv.NUM_HEADS, mv.DEPTH
        self.drop_rate = mv.DROPOUT_RATE
        self.cls_embed_on = mv.CLS_EMBED_ON
        self.sep_pos_embed = mv.SEP_POS_EMBED
        if mv.NORM != "layernorm":
            raise NotImplementedError("Only supports layernorm.")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_classes = num_classes
        self.compute_dtype = torch.bfloat16 if cfg.TRAIN.MIXED_PRECISION else torch.float32
        self.patch_embed = stem_helper.PatchEmbed(dim_in=in_chans, dim_out=embed_dim, kernel=mv.PATCH_KERNEL,
                                                  stride=mv.PATCH_STRIDE, padding=mv.PATCH_PADDING, conv_2d=False)
        self.patch_embed.compute_dtype = self.compute_dtype
        self.input_dims = [temporal_size, spatial_size, spatial_size]
        self.patch_dims = [self.input_dims[i] // self.patch_stride[i] for i in range(3)]
        num_patches = math.prod(self.patch_dims)
        dpr = [x.item() for x in torch.linspace(0, mv.DROPPATH_RATE, depth)]
        if self.cls_embed_on:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        pos_embed_dim = num_patches + 1 if self.cls_embed_on else num_patches
        if self.sep_pos_embed:
            self.pos_embed_spatial = nn.Parameter(torch.zeros(1, self.patch_dims[1] * self.patch_dims[2], embed_dim))
            self.pos_embed_temporal = nn.Parameter(torch.zeros(1, self.patch_dims[0], embed_dim))
            if self.cls_embed_on:
                self.pos_embed_class = nn.Parameter(torch.zeros(1, 1, embed_dim))
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, pos_embed_dim, embed_dim))
        if self.drop_rate > 0.0:
            self.pos_drop = nn.Dropout(p=self.drop_rate)

        dim_mul, head_mul = torch.ones(depth + 1), torch.ones(depth + 1)
        for i, m in mv.DIM_MUL:
            dim_mul[i] = m
        for i, m in mv.HEAD_MUL:
            head_mul[i] = m
        pool_q, pool_kv = [[] for _ in range(depth)], [[] for _ in range(depth)]
        stride_q, stride_kv = [[] for _ in range(depth)], [[] for _ in range(depth)]
        for e in mv.POOL_Q_STRIDE:
            stride_q[e[0]] = list(e[1:])
            pool_q[e[0]] = list(mv.POOL_KVQ_KERNEL) if mv.POOL_KVQ_KERNEL is not None else [s + 1 if s > 1 else s for s in e[1:]]
        kv_strides = mv.POOL_KV_STRIDE
        if mv.POOL_KV_STRIDE_ADAPTIVE is not None:                       # :877-888: the kv stride shrinks as q is pooled
            _stride_kv = list(mv.POOL_KV_STRIDE_ADAPTIVE)
            kv_strides = []
            for i in range(depth):
                if len(stride_q[i]) > 0:
                    _stride_kv = [max(_stride_kv[d] // stride_q[i][d], 1) for d in range(len(_stride_kv))]
                kv_strides.append([i] + _stride_kv)
        for e in (kv_strides or []):
            stride_kv[e[0]] = list(e[1:])
            pool_kv[e[0]] = list(mv.POOL_KVQ_KERNEL) if mv.POOL_KVQ_KERNEL is not None else [s + 1 if s > 1 else s for s in e[1:]]
        self.norm_stem = norm_layer(embed_dim) if mv.NORM_STEM else None

        i_num_frames = cfg.DATA.NUM_FRAMES // self.patch_stride[0]
        self.blocks = nn.ModuleList()
        self.orvit_blocks = nn.ModuleList()
        for i in range(depth):
            num_heads = round_width(num_heads, head_mul[i].item())
            embed_dim = round_width(embed_dim, dim_mul[i].item(), divisor=num_heads)
            dim_out = round_width(embed_dim, dim_mul[i + 1].item(), divisor=round_width(num_heads, head_mul[i + 1].item()))

            def orvit():
                return ORViT(cfg=cfg, dim=embed_dim, dim_out=dim_out, num_heads=num_heads, mlp_ratio=mv.MLP_RATIO,
                             qkv_bias=mv.QKV_BIAS, drop=self.drop_rate, attn_drop=self.drop_rate, drop_path=dpr[i],
                             norm_layer=norm_layer, nb_frames=i_num_frames)
            if i in cfg.ORVIT.LAYERS:
                self.blocks.append(orvit())
            else:
                self.blocks.append(MultiScaleBlock(
                    dim=embed_dim, dim_out=dim_out, num_heads=num_heads, mlp_ratio=mv.MLP_RATIO, qkv_bias=mv.QKV_BIAS,
                    drop_rate=self.drop_rate, drop_path=dpr[i], norm_layer=norm_layer, kernel_q=pool_q[i],
                    kernel_kv=pool_kv[i], stride_q=stride_q[i], stride_kv=stride_kv[i], mode=mv.MODE,
                    has_cls_embed=self.cls_embed_on, pool_first=pool_first,
                    ignore_111_kv_kernel=mv.POOL_KV_IGNORE_111_KERNEL))
            tstride = stride_q[i][0] if stride_q[i] else 1
            if i in cfg.ORVIT.ADD_LAYERS:
                assert not stride_q[i]
                self.orvit_blocks.append(orvit())
            else:
                self.orvit_blocks.append(None)
            i_num_frames //= tstride
        embed_dim = dim_out
        self.norm = norm_layer(embed_dim)
        self.head = TransformerBasicHead(embed_dim, num_classes, dropout_rate=cfg.MODEL.DROPOUT_RATE,
                                         act_func=cfg.MODEL.HEAD_ACT)
        if self.sep_pos_embed:
            trunc_normal_(self.pos_embed_spatial, std=0.02)
            trunc_normal_(self.pos_embed_temporal, std=0.02)
            if self.cls_embed_on:
                trunc_normal_(self.pos_embed_class, std=0.02)
        else:
            trunc_normal_(self.pos_embed, std=0.02)
        if self.cls_embed_on:
            trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        if not self.cfg.MVIT.ZERO_DECAY_POS_CLS:
            return {}
        if self.sep_pos_embed:
            names = {"pos_embed_spatial", "pos_embed_temporal", "pos_embed_class"}
            return names | {"cls_token"} if self.cls_embed_on else names
        return {"pos_embed", "cls_token"} if self.cls_embed_on else {"pos_embed"}

    def forward(self, x, metadata, bboxes=None):
        x = x[0]
        dev = x.device
        x = self.patch_embed(x)
        T = self.cfg.DATA.NUM_FRAMES // self.patch_stride[0]
        H = self.cfg.DATA.TRAIN_CROP_SIZE // self.patch_stride[1]
        W = self.cfg.DATA.TRAIN_CROP_SIZE // self.patch_stride[2]
        B = x.shape[0]
        dt = self.compute_dtype
        if self.cls_embed_on:
            x = torch.cat((self.cls_token.to(dt).expand(B, -1, -1), x), dim=1)
        if self.sep_pos_embed:
            pos_embed = self.pos_embed_spatial.repeat(1, self.patch_dims[0], 1) + torch.repeat_interleave(
                self.pos_embed_temporal, self.patch_dims[1] * self.patch_dims[2], dim=1)
            if self.cls_embed_on:
                pos_embed = torch.cat([self.pos_embed_class, pos_embed], 1)
            x = x + pos_embed.to(dt)
        else:
            x = x + self.pos_embed.to(dt)
        if self.drop_rate:
            x = self.pos_drop(x)
        if self.norm_stem:
            x = ops.layer_norm(x, self.norm_stem.weight, self.norm_stem.bias, self.norm_stem.eps)
        thw = [T, H, W]
        x = x.contiguous()
        for blk, blk_orvit in zip(self.blocks, self.orvit_blocks):
            x_prev, thw_prev = x, thw
            x, thw = blk(x_prev, metadata, thw_prev)
            if blk_orvit is not None:
                x_orvit, _ = blk_orvit(x_prev, metadata, thw_prev)
                x = x + x_orvit
        x = ops.layer_norm(x.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps)
        x = x[:, 0] if self.cls_embed_on else x.mean(1)
        assert x.device == dev
        return self.head(x.contiguous())
