"""oracle/focus_oracle.py -- TEST INFRASTRUCTURE, not product code.

CPU restatement (PyTorch-CPU tensor algebra, any float dtype, autograd-capable) of the reference's
object-centric video hot path.  Every function cites the reference file:line it follows
(paths relative to /root/reference).  Parameters are passed as a flat ``dict[str, Tensor]`` that uses
the reference's state_dict names, so one set of weights drives the reference module, this oracle and
the HIP product path.

Pinning: tests/test_oracle_golden.py checks every function here against fixtures emitted by
oracle/make_golden.py from the reference's own source files (executed in the build container).
The single exception is RoIAlign (third-party torchvision arithmetic, absent here):
"parity unpinned" -- see oracle/roi_align_ref.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _clib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libfocus_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C library missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


# ----------------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------------
def linear(p, name, x):
    """nn.Linear as used throughout (weight [out,in], optional bias)."""
    w = p[name + ".weight"]
    b = p.get(name + ".bias")
    y = x @ w.t()
    return y if b is None else y + b


def layer_norm(p, name, x, eps):
    """nn.LayerNorm over the last axis; eps=1e-6 in Motionformer (video_model_builder.py:1129),
    1e-5 (default) in STEVE (steve.py:35-37)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * p[name + ".weight"] + p[name + ".bias"]


def gelu(x):
    """nn.GELU() default = exact erf form (common.py:20)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mlp(p, name, x):
    """common.py:26-34 / ORViT/utils.py:92-98: fc1 -> GELU -> fc2 (dropout p=0)."""
    return linear(p, name + ".fc2", gelu(linear(p, name + ".fc1", x)))


def split_heads(x, h):
    """'b n (h d) -> b h n d' (attention.py:509-510)."""
    B, N, C = x.shape
    return x.reshape(B, N, h, C // h).permute(0, 2, 1, 3)


def merge_heads(x):
    """'b h n d -> b n (h d)'."""
    B, h, N, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, N, h * d)


# ----------------------------------------------------------------------------------------------
# trajectory attention  (attention.py:499-557)
# ----------------------------------------------------------------------------------------------
def trajectory_attention(p, name, x, thw, heads, use_original_code=True):
    """x [B, 1+F*P, C]; token order [cls | frame0 (P) | frame1 (P) | ...]."""
    B, N, C = x.shape
    Fr, P = thw[0], thw[1] * thw[2]
    S = Fr * P
    assert N == S + 1
    d = C // heads
    scale = d ** -0.5

    qkv = linear(p, name + ".qkv", x)                                   # :506
    q, k, v = (split_heads(t, heads) for t in qkv.split(C, dim=-1))      # :509-510  [B,h,N,d]

    # cls row attends over all N keys (:514-519)
    cls_logit = (q[:, :, :1] * scale) @ k.transpose(-1, -2)              # [B,h,1,N]
    cls_out = merge_heads(torch.softmax(cls_logit, dim=-1) @ v)          # [B,1,C]

    # patch rows: one softmax per (query, frame) over that frame's P keys (:524-529)
    q_, k_, v_ = q[:, :, 1:], k[:, :, 1:], v[:, :, 1:]
    logits = (q_ @ k_.transpose(-1, -2)).reshape(B, heads, S, Fr, P) * scale
    A = torch.softmax(logits, dim=-1)
    vf = v_.reshape(B, heads, Fr, P, d)
    xt = torch.einsum("bhsfp,bhfpd->bhsfd", A, vf)                       # x~  [B,h,S,F,d]

    # temporal step (:532-549)
    xt_m = xt.permute(0, 2, 3, 1, 4).reshape(B, S, Fr, C)                # [B,S,F,C]
    own = torch.arange(S) // P
    x_diag = xt_m[:, torch.arange(S), own]                               # [B,S,C]  (:533-535)
    q2 = split_heads(linear(p, name + ".proj_q", x_diag), heads) * scale  # [B,h,S,d]
    kv2 = linear(p, name + ".proj_kv", xt_m)                             # [B,S,F,2C]
    k2 = kv2[..., :C].reshape(B, S, Fr, heads, d).permute(0, 3, 1, 2, 4)  # [B,h,S,F,d]
    A2 = torch.softmax((q2.unsqueeze(3) * k2).sum(-1), dim=-1)           # [B,h,S,F]
    if use_original_code:
        val = xt                                                         # :545-547
    else:
        val = kv2[..., C:].reshape(B, S, Fr, heads, d).permute(0, 3, 1, 2, 4)
    out = merge_heads((A2.unsqueeze(-1) * val).sum(3))                   # [B,S,C]

    y = linear(p, name + ".proj", torch.cat([cls_out, out], dim=1))      # :553-556
    return y


def trajectory_block(p, name, x, thw, heads, eps=1e-6):
    """TrajectoryAttentionBlock.forward (attention.py:467-476), eval mode (DropPath = identity)."""
    x = x + trajectory_attention(p, name + ".attn", layer_norm(p, name + ".norm1", x, eps), thw, heads)
    x = x + mlp(p, name + ".mlp", layer_norm(p, name + ".norm2", x, eps))
    return x


def joint_attention_block(p, name, x, heads, eps=1e-6):
    """SeltAttentionBlock.forward (attention.py:426-432) with SelfAttention.forward (:369-385)."""
    B, N, C = x.shape
    d = C // heads
    y = layer_norm(p, name + ".norm1", x, eps)
    qkv = linear(p, name + ".attn.qkv", y)
    q, k, v = (split_heads(t, heads) for t in qkv.split(C, dim=-1))
    a = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, dim=-1)
    y = linear(p, name + ".attn.proj", merge_heads(a @ v))
    x = x + y
    x = x + mlp(p, name + ".mlp", layer_norm(p, name + ".norm2", x, eps))
    return x


# ----------------------------------------------------------------------------------------------
# RoIAlign (third-party arithmetic; restated in oracle/roi_align_ref.c)
# ----------------------------------------------------------------------------------------------
def _f32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _RoiAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, roi_img, PH, PW, scale, sampling_ratio, aligned):
        NI, C, H, W = feat.shape
        K = rois.shape[0]
        f, r = _f32(feat), _f32(rois)
        im = np.ascontiguousarray(roi_img.cpu().numpy().astype(np.int32))
        out = np.zeros((K, C, PH, PW), np.float32)
        _clib().oracle_roi_align_fwd(_ptr(f), _ptr(r), _ptr(im), _ptr(out), NI, C, H, W, K, PH, PW,
                                     ctypes.c_float(scale), int(sampling_ratio), int(aligned))
        ctx.args = (r, im, (NI, C, H, W), K, PH, PW, scale, sampling_ratio, aligned, feat.dtype)
        return torch.from_numpy(out).to(feat.dtype)

    @staticmethod
    def backward(ctx, dout):
        r, im, (NI, C, H, W), K, PH, PW, scale, sr, al, dt = ctx.args
        d = _f32(dout)
        dfeat = np.zeros((NI, C, H, W), np.float32)
        _clib().oracle_roi_align_bwd(_ptr(d), _ptr(r), _ptr(im), _ptr(dfeat), NI, C, H, W, K, PH, PW,
                                     ctypes.c_float(scale), int(sr), int(al))
        return torch.from_numpy(dfeat).to(dt), None, None, None, None, None, None, None


def roi_align(feat, rois, roi_img, output_size, spatial_scale, sampling_ratio=-1, aligned=True):
    """feat [NI,C,H,W]; rois [K,4] xyxy pixels; roi_img [K] int -> [K,C,PH,PW] (fp32 arithmetic)."""
    PH, PW = output_size
    return _RoiAlignFn.apply(feat, rois, roi_img, PH, PW, float(spatial_scale), sampling_ratio, aligned)


def roi_align_list(feat, boxes_list, output_size, spatial_scale, sampling_ratio=-1, aligned=True):
    """torchvision calling convention used at ORViT/utils.py:64-71 (list of [O,4] per image)."""
    rois = torch.cat(list(boxes_list), dim=0)
    idx = torch.cat([torch.full((b.shape[0],), i, dtype=torch.int32) for i, b in enumerate(boxes_list)])
    return roi_align(feat, rois, idx, output_size, spatial_scale, sampling_ratio, aligned)


def roi_align_indices(rois, H, W, output_size, spatial_scale, sampling_ratio=-1, aligned=True):
    """Integer side of RoIAlign: (grid [K,2], neighbours [K,PH,PW,4]) as int32 numpy arrays."""
    PH, PW = output_size
    r = _f32(rois)
    K = r.shape[0]
    grid = np.zeros((K, 2), np.int32)
    nbr = np.zeros((K, PH, PW, 4), np.int32)
    _clib().oracle_roi_align_indices(_ptr(r), _ptr(grid), _ptr(nbr), H, W, K, PH, PW,
                                     ctypes.c_float(spatial_scale), int(sampling_ratio), int(aligned))
    return grid, nbr


def cxcywh_to_xyxy(b):
    """box_ops.py:17-21."""
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def objects_crops(feat, boxes, crop_size):
    """ObjectsCrops.forward (ORViT/utils.py:48-76).  feat [B,C,T,H,W], boxes [B,T,O,4] cxcywh in [0,1]
    -> [B,O,T,C,H,W].  Boxes are un-normalised with TRAIN_CROP_SIZE (:40,62-63)."""
    B, C, T, H, W = feat.shape
    O = boxes.shape[2]
    f = feat.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    rois = cxcywh_to_xyxy(boxes.reshape(B * T, O, 4)).float() * float(crop_size)
    img = torch.arange(B * T, dtype=torch.int32).repeat_interleave(O)
    crops = roi_align(f, rois.reshape(-1, 4), img, (H, W), H / crop_size, -1, True)
    return crops.reshape(B, T, O, C, H, W).permute(0, 2, 1, 3, 4, 5)


# ----------------------------------------------------------------------------------------------
# box layout (ORViT/utils.py:8-28 -> layout.py:28-63, 98-130, 205-237)
# ----------------------------------------------------------------------------------------------
def box_layout(vecs, boxes, H, W):
    """vecs [B,T,O,C], boxes [B,T,O,4] cxcywh -> [B,T,H,W,C] (always fp32, layout.py:53).

    Per frame: drop boxes whose xyxy is all-zero; sample a constant 8x8 image per object with a
    bilinear, zero-padded, align_corners=True grid built from (lin - x0)/x1, (lin - y0)/y1 -- the
    reference reads xyxy columns 2,3 as width,height (layout.py:113-120), which is kept; sum over
    objects."""
    B, T, O, C = vecs.shape
    out = torch.zeros(B, T, H, W, C, dtype=torch.float32)
    lin_x = torch.linspace(0, 1, steps=W)
    lin_y = torch.linspace(0, 1, steps=H)
    frames = []
    for b in range(B):
        for t in range(T):
            xy = cxcywh_to_xyxy(boxes[b, t])
            keep = (xy != 0).any(dim=-1)
            acc = torch.zeros(H, W, C, dtype=torch.float32)
            for o in range(O):
                if not bool(keep[o]):
                    continue
                x0, y0, ww, hh = xy[o]
                gx = ((lin_x - x0) / ww) * 2 - 1
                gy = ((lin_y - y0) / hh) * 2 - 1
                grid = torch.stack([gx.view(1, W).expand(H, W), gy.view(H, 1).expand(H, W)], dim=-1)
                img = vecs[b, t, o].float().view(1, C, 1, 1).expand(1, C, 8, 8)
                s = F.grid_sample(img, grid.unsqueeze(0).float(), mode="bilinear", padding_mode="zeros",
                                  align_corners=True)                    # [1,C,H,W]
                acc = acc + s[0].permute(1, 2, 0)
            frames.append(acc)
    return torch.stack(frames).reshape(B, T, H, W, C)


def box_layout_weights(boxes, H, W):
    """Closed form of the layout sampling weights:  w(u)=clamp(min(7u+1, 8-7u),0,1).
    Returns wy [B,T,O,H], wx [B,T,O,W], keep [B,T,O]."""
    xy = cxcywh_to_xyxy(boxes.float())
    keep = (xy != 0).any(dim=-1)
    lin_x = torch.linspace(0, 1, steps=W)
    lin_y = torch.linspace(0, 1, steps=H)

    def w(u):
        return torch.clamp(torch.minimum(7 * u + 1, 8 - 7 * u), 0, 1)

    wx = w((lin_x - xy[..., 0:1]) / xy[..., 2:3])
    wy = w((lin_y - xy[..., 1:2]) / xy[..., 3:4])
    zero = torch.zeros(())
    wx = torch.where(keep[..., None], wx, zero)      # dropped boxes would give 0/0 here
    wy = torch.where(keep[..., None], wy, zero)
    return wy, wx, keep


# ----------------------------------------------------------------------------------------------
# ORViT block (orvit.py:116-172) and motion stream (orvit.py:204-269)
# ----------------------------------------------------------------------------------------------
def relu_pair(p, name, x):
    """nn.Sequential(Linear(no bias), ReLU, Linear(no bias), ReLU) (orvit.py:59-64, 67-72, 225-230)."""
    return torch.relu(torch.relu(x @ p[name + ".0.weight"].t()) @ p[name + ".2.weight"].t())


def motion_stream(p, name, boxes, H, W, heads, eps=1e-6):
    """MotionStream.forward (orvit.py:254-269), joint attention type, no separate pos-emb."""
    B, T, O, _ = boxes.shape
    e = relu_pair(p, name + ".c_coord_to_feature", boxes) + p[name + ".box_categories"]
    C = e.shape[-1]
    e = joint_attention_block(p, name + ".attn", e.reshape(B, T * O, C), heads, eps).reshape(B, T, O, C)
    return box_layout(e, boxes, H, W).reshape(B, T * H * W, C)


def orvit_block(p, name, x, boxes_in, thw, heads, crop_size, eps=1e-6, with_motion_stream=True):
    """ORViT.forward.  x [B,1+T*H*W,C]; boxes_in [B,T_in,O,4] cxcywh (metadata['orvit_bboxes'])."""
    B, N, C = x.shape
    T, H, W = thw
    boxes = boxes_in[:, :: boxes_in.shape[1] // T]                        # :131-132
    O = boxes.shape[2]
    cls, pt = x[:, :1], x[:, 1:]
    feat = pt.permute(0, 2, 1).reshape(B, C, T, H, W)                     # :126
    crops = objects_crops(feat, boxes, crop_size).to(x.dtype)             # [B,O,T,C,H,W]  :135
    obj = relu_pair(p, name + ".patch_to_d", crops.permute(0, 1, 2, 4, 5, 3))
    obj = obj.amax(dim=(-3, -2)).permute(0, 2, 1, 3)                      # [B,T,O,C]  :138-139
    obj = obj + p[name + ".box_categories"] + relu_pair(p, name + ".c_coord_to_feature", boxes)
    tokens = torch.cat([pt.reshape(B, T, H * W, C), obj], dim=2).reshape(B, T * (H * W + O), C)
    tokens = torch.cat([cls, tokens], dim=1)                              # :145-147
    a = trajectory_attention(p, name + ".attn", layer_norm(p, name + ".norm1", tokens, eps),
                             [T, H * W + O, 1], heads)                    # :149-152
    cls2, a = a[:, :1], a[:, 1:]
    a = a.reshape(B, T, H * W + O, C)[:, :, : H * W].reshape(B, T * H * W, C)   # :157
    if with_motion_stream:
        m = motion_stream(p, name + ".motion_stream", boxes, H, W, heads, eps).to(x.dtype)
        a = a + mlp(p, name + ".motion_mlp", m)                           # :161-163
    x = x + torch.cat([cls2, a], dim=1)                                   # :169
    x = x + mlp(p, name + ".mlp", layer_norm(p, name + ".norm2", x, eps))  # :170
    return x


# ----------------------------------------------------------------------------------------------
# Motionformer (video_model_builder.py:1271-1353, stem_helper.py:317-320)
# ----------------------------------------------------------------------------------------------
def patch_embed(p, name, x, kernel):
    """Conv3d with kernel == stride, no padding, then tokens in (t,h,w) order."""
    y = F.conv3d(x, p[name + ".proj.weight"], p[name + ".proj.bias"], stride=kernel)
    return y.flatten(2).transpose(1, 2)


def motionformer_embed(p, x, temporal_resolution, kernel, crop_size):
    """forward_features up to the blocks (:1271-1322), POS_EMBED == 'separate'."""
    tok = patch_embed(p, "patch_embed_3d", x, kernel)
    B, S, C = tok.shape
    npatch = S // temporal_resolution
    pos = p["pos_embed"]
    if crop_size != 224:                                                  # :1285-1300
        n0 = pos.shape[1] - 1
        g = int(math.sqrt(n0))
        sp = pos[:, 1:].reshape(1, g, g, C).permute(0, 3, 1, 2)
        sp = F.interpolate(sp, scale_factor=math.sqrt(npatch / n0), mode="bicubic")
        spatial = sp.permute(0, 2, 3, 1).reshape(1, -1, C)
    else:
        spatial = pos[:, 1:]
    total = spatial.repeat(1, temporal_resolution, 1) + p["temp_embed"].repeat_interleave(npatch, 1)
    total = torch.cat([pos[:, :1], total], dim=1)                         # :1306-1315
    xx = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + total
    return xx, npatch


def motionformer_forward(p, x, boxes, cfg, training=True):
    """Motionformer.forward (video_model_builder.py:1338-1353); cfg is a dict with keys depth, heads, orvit_layers,
    temporal_resolution, patch (t,h,w), crop.  With EPIC-Kitchens heads in `p` (head0 / head1, :1229-1231) the return
    value is the reference's (verb, {'verb', 'noun'}) pair (:1341-1348), else the single-head logits."""
    xx, npatch = motionformer_embed(p, x, cfg["temporal_resolution"], cfg["patch"], cfg["crop"])
    side = int(npatch ** 0.5)
    thw = [cfg["temporal_resolution"], side, side]
    for i in range(cfg["depth"]):
        name = "blocks.%d" % i
        if i in cfg["orvit_layers"]:
            xx = orvit_block(p, name, xx, boxes, thw, cfg["heads"], cfg["crop"])
        else:
            xx = trajectory_block(p, name, xx, thw, cfg["heads"])
    feat = layer_norm(p, "norm", xx, 1e-6)[:, 0]
    feat = torch.tanh(linear(p, "pre_logits.fc", feat))                   # HEAD_ACT tanh, USE_MLP
    if "head0.weight" in p:
        outs = [linear(p, "head%d" % i, feat) for i in range(2)]
        if not training:
            outs = [torch.softmax(o, dim=-1) for o in outs]
        return outs[0], {"verb": outs[0], "noun": outs[1]}
    logits = linear(p, "head", feat)
    return logits if training else torch.softmax(logits, dim=-1)


# ----------------------------------------------------------------------------------------------
# MViT with ORViT blocks (video_model_builder.py:765-1101, attention.py:16-352, utils.py:31-44, head_helper.py:363-419)
# ----------------------------------------------------------------------------------------------
def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast/models/utils.py:31-44."""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def mvit_plan(cfg):
    """The per-layer (dim, dim_out, heads, kernel_q, stride_q, kernel_kv, stride_kv) of MViT.__init__ (:857-913).  cfg: dict
    with embed_dim, heads, depth, dim_mul, head_mul ([[layer, factor], ...]), pool_q_stride ([[layer, t, h, w], ...]),
    pool_kvq_kernel, pool_kv_stride_adaptive."""
    depth = cfg["depth"]
    dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
    for i, m in cfg["dim_mul"]:
        dim_mul[i] = m
    for i, m in cfg["head_mul"]:
        head_mul[i] = m
    stride_q = [[] for _ in range(depth)]
    for e in cfg["pool_q_stride"]:
        stride_q[e[0]] = list(e[1:])
    kern = list(cfg["pool_kvq_kernel"])
    stride_kv, skv = [], list(cfg["pool_kv_stride_adaptive"])
    for i in range(depth):                                                # :877-888
        if stride_q[i]:
            skv = [max(skv[d] // stride_q[i][d], 1) for d in range(3)]
        stride_kv.append(list(skv))
    plan, dim, heads = [], cfg["embed_dim"], cfg["heads"]
    for i in range(depth):
        heads = round_width(heads, head_mul[i])
        dim = round_width(dim, dim_mul[i], divisor=heads)
        dim_out = round_width(dim, dim_mul[i + 1], divisor=round_width(heads, head_mul[i + 1]))
        plan.append(dict(dim=dim, dim_out=dim_out, heads=heads, kernel_q=kern if stride_q[i] else [], stride_q=stride_q[i],
                         kernel_kv=kern, stride_kv=stride_kv[i]))
    return plan


def _pool_tokens(t, thw, pool, has_cls=True):
    """attention_pool (attention.py:16-50) around `pool`: a function of the [B*heads, C, T, H, W] grid."""
    cls, t = (t[:, :, :1], t[:, :, 1:]) if has_cls else (None, t)
    B, N, L, C = t.shape
    T, H, W = thw
    g = pool(t.reshape(B * N, T, H, W, C).permute(0, 4, 1, 2, 3))
    thw2 = [g.shape[2], g.shape[3], g.shape[4]]
    t = g.reshape(B, N, C, -1).transpose(2, 3)
    if has_cls:
        t = torch.cat([cls, t], dim=2)
    return t, thw2


def multiscale_block(p, name, x, thw, lay, eps=1e-6):
    """MultiScaleBlock.forward (attention.py:343-352) with MultiScaleAttention.forward (:158-258), mode 'conv', pooling after
    the projections."""
    B, N, C = x.shape
    h = lay["heads"]
    y = layer_norm(p, name + ".norm1", x, eps)
    qkv = {}
    for n in "qkv":
        qkv[n] = split_heads(linear(p, name + ".attn." + n, y), h)        # [B,h,N,d]
    shapes = {}
    for n, kernel, stride in (("q", lay["kernel_q"], lay["stride_q"]), ("k", lay["kernel_kv"], lay["stride_kv"]),
                              ("v", lay["kernel_kv"], lay["stride_kv"])):
        shapes[n] = thw
        if len(kernel) > 0 and not (all(a == 1 for a in kernel) and all(a == 1 for a in stride)):
            w = p[name + ".attn.pool_%s.weight" % n]
            pad = [a // 2 for a in kernel]
            t, shapes[n] = _pool_tokens(qkv[n], thw, lambda g_: F.conv3d(g_, w, None, stride, pad, 1, w.shape[0]))
            qkv[n] = layer_norm(p, name + ".attn.norm_" + n, t, 1e-5)     # nn.LayerNorm default eps (attention.py:293)
    d = C // h
    att = torch.softmax((qkv["q"] @ qkv["k"].transpose(-1, -2)) * d ** -0.5, dim=-1)
    a = linear(p, name + ".attn.proj", merge_heads(att @ qkv["v"]))
    sq = lay["stride_q"]
    if len(sq) > 0:                                                       # pool_skip: MaxPool3d (:333-340)
        ks = [s_ + 1 if s_ > 1 else s_ for s_ in sq]
        res, _ = _pool_tokens(x.unsqueeze(1), thw, lambda g_: F.max_pool3d(g_, ks, sq, [k_ // 2 for k_ in ks]))
        res = res.squeeze(1)
    else:
        res = x
    x = res + a
    xn = layer_norm(p, name + ".norm2", x, eps)
    m = mlp(p, name + ".mlp", xn)
    if lay["dim"] != lay["dim_out"]:
        x = linear(p, name + ".proj", xn)
    return x + m, shapes["q"]


def mvit_forward(p, x, boxes, cfg, training=True):
    """MViT.forward (video_model_builder.py:1043-1101): conv stem, cls token, separate position embeddings, the blocks with
    ORViT in place of (orvit_layers) or beside (orvit_add_layers) a MultiScaleBlock, norm, cls row, TransformerBasicHead.
    cfg: mvit_plan's keys + patch_kernel / patch_stride / patch_padding, crop, frames, orvit_layers, orvit_add_layers."""
    y = F.conv3d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg["patch_stride"], cfg["patch_padding"])
    tok = y.flatten(2).transpose(1, 2)
    B = tok.shape[0]
    dims = [cfg["frames"] // cfg["patch_stride"][0], cfg["crop"] // cfg["patch_stride"][1], cfg["crop"] // cfg["patch_stride"][2]]
    pos = p["pos_embed_spatial"].repeat(1, dims[0], 1) + torch.repeat_interleave(p["pos_embed_temporal"], dims[1] * dims[2], dim=1)
    pos = torch.cat([p["pos_embed_class"], pos], 1)
    xx = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + pos
    thw = dims
    for i, lay in enumerate(mvit_plan(cfg)):
        prev, thw_prev = xx, thw
        if i in cfg["orvit_layers"]:
            xx = orvit_block(p, "blocks.%d" % i, prev, boxes, thw_prev, lay["heads"], cfg["crop"])
        else:
            xx, thw = multiscale_block(p, "blocks.%d" % i, prev, thw_prev, lay)
        if i in cfg["orvit_add_layers"]:
            xx = xx + orvit_block(p, "orvit_blocks.%d" % i, prev, boxes, thw_prev, lay["heads"], cfg["crop"])
    feat = layer_norm(p, "norm", xx, 1e-6)[:, 0]
    logits = linear(p, "head.projection", feat)
    return logits if training else torch.softmax(logits, dim=1)


def label_smoothing_ce(logits, target, smoothing=0.1):
    """losses.py:53-59."""
    lp = torch.log_softmax(logits, dim=-1)
    nll = -lp.gather(-1, target.unsqueeze(1)).squeeze(1)
    return ((1.0 - smoothing) * nll + smoothing * (-lp.mean(-1))).mean()


def ek_loss(extra_preds, labels, smoothing=0.1):
    """EKLoss (losses.py:62-95) with ce_type 'label_smoothing' + the sum the train loop forms (train_net.py:95-97)."""
    return {"verb_loss": label_smoothing_ce(extra_preds["verb"], labels["verb"], smoothing),
            "noun_loss": label_smoothing_ce(extra_preds["noun"], labels["noun"], smoothing)}


# ----------------------------------------------------------------------------------------------
# STEVE slot attention over video (steve.py:52-105, utils.py:107-118, transformer.py:4-114)
# ----------------------------------------------------------------------------------------------
def gru_cell(p, name, x, h):
    """nn.GRUCell, gate order (r,z,n) in weight_ih/weight_hh [3D,D]."""
    D = h.shape[-1]
    gi = x @ p[name + ".weight_ih"].t() + p[name + ".bias_ih"]
    gh = h @ p[name + ".weight_hh"].t() + p[name + ".bias_hh"]
    r = torch.sigmoid(gi[..., :D] + gh[..., :D])
    z = torch.sigmoid(gi[..., D:2 * D] + gh[..., D:2 * D])
    n = torch.tanh(gi[..., 2 * D:] + r * gh[..., 2 * D:])
    return (1 - z) * n + z * h


def slot_mha(p, name, x, heads):
    """MultiHeadAttention.forward (transformer.py:23-49), self-attention, no mask, dropout off."""
    B, K, D = x.shape
    d = D // heads
    q = split_heads(x @ p[name + ".proj_q.weight"].t(), heads) * d ** -0.5
    k = split_heads(x @ p[name + ".proj_k.weight"].t(), heads)
    v = split_heads(x @ p[name + ".proj_v.weight"].t(), heads)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    return merge_heads(a @ v) @ p[name + ".proj_o.weight"].t()


def slot_predictor(p, name, x, heads, num_blocks, eps=1e-5):
    """TransformerEncoder.forward (transformer.py:105-114); block 0 is_first (:75-78)."""
    for j in range(num_blocks):
        bn = "%s.blocks.%d" % (name, j)
        if j == 0:
            x = layer_norm(p, bn + ".attn_layer_norm", x, eps)
            x = x + slot_mha(p, bn + ".attn", x, heads)
        else:
            x = x + slot_mha(p, bn + ".attn", layer_norm(p, bn + ".attn_layer_norm", x, eps), heads)
        y = layer_norm(p, bn + ".ffn_layer_norm", x, eps)
        y = torch.relu(linear(p, bn + ".ffn.0", y))
        x = x + linear(p, bn + ".ffn.2", y)
    return layer_norm(p, name + ".layer_norm", x, eps)


def slot_attention_video(p, inputs, noise, num_iterations, pred_heads, pred_blocks, epsilon=1e-8,
                         eps_ln=1e-5, name=""):
    """SlotAttentionVideo.forward.  inputs [B,T,N,Din]; noise [B,K,Ds] is the N(0,1) draw the
    reference makes inside forward (steve.py:56) and is an explicit input here.
    Returns (slots [B,T,K,Ds], attn_vis [B,T,N,K])."""
    pre = name + "." if name else ""
    B, T, N, _ = inputs.shape
    Ds = noise.shape[-1]
    slots = p[pre + "slot_mu"] + torch.exp(p[pre + "slot_log_sigma"]) * noise
    x = layer_norm(p, pre + "norm_inputs", inputs, eps_ln)
    k = (x @ p[pre + "project_k.weight"].t()) * Ds ** -0.5
    v = x @ p[pre + "project_v.weight"].t()
    slots_out, attn_out = [], []
    for t in range(T):
        for i in range(num_iterations):
            prev = slots
            q = layer_norm(p, pre + "norm_slots", slots, eps_ln) @ p[pre + "project_q.weight"].t()
            a_vis = torch.softmax(k[:, t] @ q.transpose(-1, -2), dim=-1)            # [B,N,K] over slots
            a = a_vis + epsilon
            a = a / a.sum(dim=-2, keepdim=True)
            upd = a.transpose(-1, -2) @ v[:, t]                                       # [B,K,Ds]
            slots = gru_cell(p, pre + "gru", upd, prev)
            if i < num_iterations - 1:
                y = layer_norm(p, pre + "norm_mlp", slots, eps_ln)
                slots = slots + linear(p, pre + "mlp.2", torch.relu(linear(p, pre + "mlp.0", y)))
        slots_out.append(slots)
        attn_out.append(a_vis)
        slots = slot_predictor(p, pre + "predictor", slots, pred_heads, pred_blocks, eps_ln)
    return torch.stack(slots_out, dim=1), torch.stack(attn_out, dim=1)


# ----------------------------------------------------------------------------------------------
# STEVE.forward (steve.py:253-330): dVAE tokens, CNN encoder, slot attention, autoregressive decoder
# ----------------------------------------------------------------------------------------------
def _conv(p, name, x, stride=1, padding=0):
    return torch.nn.functional.conv2d(x, p[name + ".weight"], p.get(name + ".bias"), stride, padding)


def _conv_block(p, name, x, stride=1, padding=0):
    """Conv2dBlock (STEVE/utils.py:82-92): conv named `.m` + ReLU."""
    return torch.relu(_conv(p, name + ".m", x, stride, padding))


def dvae_encoder(p, x, name="dvae.encoder"):
    """dvae.py:9-18: 4x4/stride-4 conv, six 1x1 conv blocks, 1x1 conv to the vocabulary."""
    x = _conv_block(p, name + ".0", x, 4)
    for i in range(1, 7):
        x = _conv_block(p, name + ".%d" % i, x)
    return _conv(p, name + ".7", x)


def dvae_decoder(p, z, name="dvae.decoder"):
    """dvae.py:20-32."""
    ps = torch.nn.functional.pixel_shuffle
    x = _conv_block(p, name + ".0", z)
    x = _conv_block(p, name + ".1", x, 1, 1)
    x = _conv_block(p, name + ".2", x)
    x = _conv_block(p, name + ".3", x)
    x = ps(_conv_block(p, name + ".4", x), 2)
    x = _conv_block(p, name + ".6", x, 1, 1)
    x = _conv_block(p, name + ".7", x)
    x = _conv_block(p, name + ".8", x)
    x = ps(_conv_block(p, name + ".9", x), 2)
    return _conv(p, name + ".11", x)


def gumbel_softmax(logits, tau, hard, dim, noise):
    """STEVE/utils.py:47-61 with the Exp(1) draw as an explicit input."""
    eps = torch.finfo(logits.dtype).tiny
    g = (logits - (noise + eps).log()) / tau
    y_soft = torch.softmax(g, dim)
    if hard:
        index = y_soft.argmax(dim, keepdim=True)
        y_hard = torch.zeros_like(logits).scatter_(dim, index, 1.0)
        return y_hard - y_soft.detach() + y_soft
    return y_soft


def decoder_mha(p, name, q_in, kv_in, heads, causal):
    """MultiHeadAttention.forward (transformer.py:23-49) with the decoder's triu(1) mask (:126-127), dropout off."""
    d = q_in.shape[-1] // heads
    q = split_heads(q_in @ p[name + ".proj_q.weight"].t(), heads) * d ** -0.5
    k = split_heads(kv_in @ p[name + ".proj_k.weight"].t(), heads)
    v = split_heads(kv_in @ p[name + ".proj_v.weight"].t(), heads)
    a = q @ k.transpose(-1, -2)
    if causal:
        T = a.shape[-1]
        a = a.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1), float("-inf"))
    a = torch.softmax(a, dim=-1)
    return merge_heads(a @ v) @ p[name + ".proj_o.weight"].t()


def transformer_decoder(p, name, x, enc, heads, num_blocks, eps=1e-5):
    """TransformerDecoder.forward (transformer.py:146-166, :185-193); block 0 is_first."""
    for j in range(num_blocks):
        bn = "%s.blocks.%d" % (name, j)
        if j == 0:
            x = layer_norm(p, bn + ".self_attn_layer_norm", x, eps)
            x = x + decoder_mha(p, bn + ".self_attn", x, x, heads, True)
        else:
            y = layer_norm(p, bn + ".self_attn_layer_norm", x, eps)
            x = x + decoder_mha(p, bn + ".self_attn", y, y, heads, True)
        y = layer_norm(p, bn + ".encoder_decoder_attn_layer_norm", x, eps)
        x = x + decoder_mha(p, bn + ".encoder_decoder_attn", y, enc, heads, False)
        y = layer_norm(p, bn + ".ffn_layer_norm", x, eps)
        x = x + linear(p, bn + ".ffn.2", torch.relu(linear(p, bn + ".ffn.0", y)))
    return layer_norm(p, name + ".layer_norm", x, eps)


def steve_forward(p, video, tau, hard, noise, cfg):
    """STEVE.forward (steve.py:253-330), eval-mode dropout.  video [B,T,C,H,W]; noise = dict(gumbel_soft, gumbel_hard
    [B*T,vocab,H/4,W/4] Exp(1) draws, slots [B,K,Ds] N(0,1) draw); cfg = dict(img_size, num_slots, num_iters,
    pred_heads, pred_blocks, dec_heads, dec_blocks).  Returns (recon clamped, cross_entropy, dvae_mse, attns)."""
    B, T, C, H, W = video.shape
    vf = video.flatten(end_dim=1)
    z_logits = torch.log_softmax(dvae_encoder(p, vf), dim=1)                                   # :262
    z_soft = gumbel_softmax(z_logits, tau, hard, 1, noise["gumbel_soft"])                      # :263
    z_hard = gumbel_softmax(z_logits, tau, True, 1, noise["gumbel_hard"]).detach()             # :264
    z_hard = z_hard.permute(0, 2, 3, 1).flatten(start_dim=1, end_dim=2)
    z_emb = p["steve_decoder.dict.dictionary.weight"][torch.argmax(z_hard, dim=-1)]            # :266 (OneHotDictionary)
    z_emb = torch.cat([p["steve_decoder.bos"].expand(B * T, -1, -1), z_emb], dim=1)
    z_emb = z_emb + p["steve_decoder.pos.pe"][:, :z_emb.shape[1]]                              # :268
    recon = dvae_decoder(p, z_soft).reshape(B, T, C, H, W)                                     # :271
    mse = ((video - recon) ** 2).sum() / (B * T)
    # CNN encoder (steve.py:162-174: stride 1 at 64 px, else 2) + cartesian position embedding (:125-145)
    s0 = 1 if cfg["img_size"] == 64 else 2
    e = _conv_block(p, "steve_encoder.cnn.fenc.0", vf, s0, 2)
    e = _conv_block(p, "steve_encoder.cnn.fenc.1", e, 1, 2)
    e = _conv_block(p, "steve_encoder.cnn.fenc.2", e, 1, 2)
    e = _conv(p, "steve_encoder.cnn.fenc.3", e, 1, 2)
    e = e + _conv(p, "steve_encoder.pos.projection", p["steve_encoder.pos.pe"])
    He, We = e.shape[-2:]
    es = e.permute(0, 2, 3, 1).flatten(start_dim=1, end_dim=2)
    es = layer_norm(p, "steve_encoder.layer_norm", es, 1e-5)
    es = linear(p, "steve_encoder.mlp.2", torch.relu(linear(p, "steve_encoder.mlp.0", es)))
    es = es.reshape(B, T, He * We, -1)
    slots, attns = slot_attention_video(p, es, noise["slots"], cfg["num_iters"], cfg["pred_heads"], cfg["pred_blocks"],
                                        name="steve_encoder.savi")
    K = cfg["num_slots"]
    attns = attns.transpose(-1, -2).reshape(B, T, K, 1, He, We).repeat_interleave(H // He, dim=-2) \
        .repeat_interleave(W // We, dim=-1)
    attns = video.unsqueeze(2) * attns + (1.0 - attns)                                         # :300
    slots = slots @ p["steve_encoder.slot_proj.weight"].t()                                    # :303
    pred = transformer_decoder(p, "steve_decoder.tf", z_emb[:, :-1], slots.flatten(end_dim=1), cfg["dec_heads"],
                               cfg["dec_blocks"])
    pred = pred @ p["steve_decoder.head.weight"].t()
    ce = -(z_hard * torch.log_softmax(pred, dim=-1)).sum() / (B * T)                           # :306
    return recon.clamp(0.0, 1.0), ce, mse, attns
