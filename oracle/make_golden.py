"""oracle/make_golden.py -- TEST INFRASTRUCTURE.  Emits tests/golden/*.npz from the reference.

Run in the build container only (needs /root/reference):   python -m oracle.make_golden
Each fixture = seeded inputs + the reference module's state_dict + the reference module's outputs
(and input/parameter gradients for a fixed random cotangent).  Fixtures are data only; no reference
source travels.  RoIAlign inside ORViT is bound to oracle.focus_oracle.roi_align_list (torchvision is
absent: that one op stays "parity unpinned").
"""
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import focus_oracle as fo            # noqa: E402
from oracle._ref_loader import load_reference, load_motionformer  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _roi_align_tv(features, boxes, output_size, spatial_scale, sampling_ratio, aligned):
    return fo.roi_align_list(features, boxes, output_size, spatial_scale, sampling_ratio, aligned)


def ns(**kw):
    return types.SimpleNamespace(**kw)


def pack(prefix, sd):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def grads_of(module, names):
    return {"grad." + n: dict(module.named_parameters())[n].grad.detach().numpy() for n in names}


def small_cfg(crop=64, O=3, T=2):
    return ns(
        ORVIT=ns(ENABLE=True, O=O, LAYERS=[1], USE_MOTION_STREAM=True, MOTION_STREAM_ATTN_TYPE="joint",
                 MOTION_STREAM_DIM=-1, MOTION_STREAM_SEP_POS_EMB=False, INIT_WEIGHTS=False,
                 ZERO_INIT_ORVIT=False),
        DATA=ns(TRAIN_CROP_SIZE=crop, NUM_FRAMES=2 * T),
        MF=ns(PATCH_SIZE=16, PATCH_SIZE_TEMP=2, CHANNELS=3, EMBED_DIM=64, DEPTH=3, NUM_HEADS=4, MLP_RATIO=4,
              QKV_BIAS=True, DROP=0.0, DROP_PATH=0.0, HEAD_DROPOUT=0.0, VIDEO_INPUT=True,
              TEMPORAL_RESOLUTION=T, USE_MLP=True, ATTN_DROPOUT=0.0, HEAD_ACT="tanh", POS_DROPOUT=0.0,
              POS_EMBED="separate"),
        TRAIN=ns(DATASET="Ssv2"),
        MODEL=ns(NUM_CLASSES=10),
    )


def make_boxes(g, B, T, O, zero=((1, None, 2),)):
    """cxcywh in [0,1], boxes kept inside the frame; selected (b, t|None, o) slots zeroed."""
    wh = 0.1 + 0.4 * torch.rand(B, T, O, 2, generator=g)
    c = 0.3 + 0.4 * torch.rand(B, T, O, 2, generator=g)
    c = torch.minimum(torch.maximum(c, wh / 2), 1 - wh / 2)
    bx = torch.cat([c, wh], dim=-1)
    for (b, t, o) in zero:
        if t is None:
            bx[b, :, o] = 0
        else:
            bx[b, t, o] = 0
    return bx


def randomize(module, g, std=0.2):
    """Reference init leaves several tensors at zero (box_categories, conv weight); fixtures use
    seeded non-degenerate values everywhere instead (weights are part of the fixture)."""
    with torch.no_grad():
        for n, prm in module.named_parameters():
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or "norm" in n and n.endswith(".weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=g))
            else:
                prm.copy_(std * torch.randn(prm.shape, generator=g))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(True)
    mods = load_reference(_roi_align_tv)
    att, orv, outl = mods["attention"], mods["orvit"], mods["orvit_utils"]
    g = torch.Generator().manual_seed(1234)

    # 1. trajectory attention (attention.py:479-557)
    m = att.TrajectoryAttention(64, num_heads=4, qkv_bias=True).double()
    randomize(m, g)
    x = torch.randn(2, 1 + 2 * 16, 64, generator=g, dtype=torch.float64, requires_grad=True)
    ct = torch.randn(2, 33, 64, generator=g, dtype=torch.float64)
    y, _ = m(x, [2, 4, 4])
    (y * ct).sum().backward()
    np.savez(os.path.join(OUT, "traj_attn.npz"), x=x.detach().numpy(), ct=ct.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), thw=np.array([2, 4, 4]), heads=4, **pack("p.", m.state_dict()),
             **grads_of(m, ["qkv.weight", "qkv.bias", "proj_q.weight", "proj_kv.weight", "proj_kv.bias",
                            "proj.weight"]))

    # 1b. non-square frame size P=18+3 (ORViT calling convention thw=[T, HW+O, 1])
    m = att.TrajectoryAttention(32, num_heads=2, qkv_bias=True).double()
    randomize(m, g)
    x = torch.randn(1, 1 + 3 * 21, 32, generator=g, dtype=torch.float64, requires_grad=True)
    ct = torch.randn(1, 64, 32, generator=g, dtype=torch.float64)
    y, _ = m(x, [3, 21, 1])
    (y * ct).sum().backward()
    np.savez(os.path.join(OUT, "traj_attn_p21.npz"), x=x.detach().numpy(), ct=ct.numpy(),
             y=y.detach().numpy(), dx=x.grad.numpy(), thw=np.array([3, 21, 1]), heads=2,
             **pack("p.", m.state_dict()), **grads_of(m, ["qkv.weight", "proj_kv.weight"]))

    # 2. trajectory block + 3. joint (motion-stream) block
    from functools import partial
    ln = partial(torch.nn.LayerNorm, eps=1e-6)
    m = att.TrajectoryAttentionBlock(dim=64, num_heads=4, qkv_bias=True, norm_layer=ln).double()
    randomize(m, g)
    x = torch.randn(2, 33, 64, generator=g, dtype=torch.float64, requires_grad=True)
    ct = torch.randn(2, 33, 64, generator=g, dtype=torch.float64)
    y, _ = m(x, None, [2, 4, 4])
    (y * ct).sum().backward()
    np.savez(os.path.join(OUT, "traj_block.npz"), x=x.detach().numpy(), ct=ct.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), thw=np.array([2, 4, 4]), heads=4, **pack("p.", m.state_dict()),
             **grads_of(m, ["norm1.weight", "norm1.bias", "mlp.fc1.weight", "mlp.fc2.bias"]))

    m = att.SeltAttentionBlock(dim=64, num_heads=4, qkv_bias=True, norm_layer=ln).double()
    randomize(m, g)
    x = torch.randn(2, 6, 64, generator=g, dtype=torch.float64, requires_grad=True)
    ct = torch.randn(2, 6, 64, generator=g, dtype=torch.float64)
    y, _ = m(x, None, None)
    (y * ct).sum().backward()
    np.savez(os.path.join(OUT, "joint_block.npz"), x=x.detach().numpy(), ct=ct.numpy(), y=y.detach().numpy(),
             dx=x.grad.numpy(), heads=4, **pack("p.", m.state_dict()),
             **grads_of(m, ["attn.qkv.weight", "attn.proj.bias"]))

    # 4. box layout (ORViT/utils.py:8-28 + layout.py) incl. a zero box and a border-touching box
    B, T, O, C, H, W = 2, 2, 3, 8, 5, 5
    boxes = make_boxes(g, B, T, O, zero=((1, 0, 2), (0, None, 1)))
    boxes[0, 0, 0] = torch.tensor([0.25, 0.25, 0.5, 0.5])       # touches the top-left corner
    vecs = torch.randn(B, T, O, C, generator=g, requires_grad=True)
    ct = torch.randn(B, C, T, H, W, generator=g)
    lay = outl.box2spatial_layout(boxes, vecs, H, W)              # [B,C,T,H,W]
    (lay * ct).sum().backward()
    np.savez(os.path.join(OUT, "box_layout.npz"), boxes=boxes.numpy(), vecs=vecs.detach().numpy(),
             ct=ct.numpy(), out=lay.detach().numpy(), dvecs=vecs.grad.numpy())

    # 5. ORViT block (orvit.py:39-172), fp32 (layout and RoIAlign force fp32 inside)
    cfg = small_cfg()
    m = orv.ORViT(cfg=cfg, dim=64, num_heads=4, mlp_ratio=4, qkv_bias=True, norm_layer=ln, nb_frames=2)
    randomize(m, g)
    boxes = make_boxes(g, 2, 4, 3)                                # T_in = 4 -> every 2nd frame is used
    x = torch.randn(2, 1 + 2 * 16, 64, generator=g, requires_grad=True)
    ct = torch.randn(2, 33, 64, generator=g)
    y, _ = m(x, {"orvit_bboxes": boxes.clone()}, [2, 4, 4])
    (y * ct).sum().backward()
    np.savez(os.path.join(OUT, "orvit_block.npz"), x=x.detach().numpy(), boxes=boxes.numpy(), ct=ct.numpy(),
             y=y.detach().numpy(), dx=x.grad.numpy(), thw=np.array([2, 4, 4]), heads=4, crop=64,
             **pack("p.", m.state_dict()),
             **grads_of(m, ["patch_to_d.0.weight", "box_categories", "c_coord_to_feature.2.weight",
                            "motion_stream.box_categories", "motion_stream.attn.attn.qkv.weight",
                            "motion_mlp.fc1.weight", "attn.qkv.weight", "norm1.weight"]))

    # 6. slot attention over video (steve.py:11-105), dropout 0 (SLOTS.PREDICTOR_DROPOUT default)
    st = mods["steve"]
    m = st.SlotAttentionVideo(num_iterations=3, num_slots=3, input_size=12, slot_size=8, mlp_hidden_size=16,
                              num_predictor_blocks=2, num_predictor_heads=2, dropout=0.0).double()
    randomize(m, g, std=0.4)
    inp = torch.randn(2, 3, 16, 12, generator=g, dtype=torch.float64, requires_grad=True)
    torch.manual_seed(77)
    noise = torch.empty(2, 3, 8, dtype=torch.float64).normal_()   # the draw forward() makes at steve.py:56
    torch.manual_seed(77)
    slots, attns = m(inp)
    cs = torch.randn(slots.shape, generator=g, dtype=torch.float64)
    ca = torch.randn(attns.shape, generator=g, dtype=torch.float64)
    ((slots * cs).sum() + (attns * ca).sum()).backward()
    np.savez(os.path.join(OUT, "slot_attention.npz"), inputs=inp.detach().numpy(), noise=noise.numpy(),
             slots=slots.detach().numpy(), attns=attns.detach().numpy(), ct_slots=cs.numpy(),
             ct_attns=ca.numpy(), dinputs=inp.grad.numpy(), iters=3, pred_heads=2, pred_blocks=2,
             **pack("p.", m.state_dict()),
             **grads_of(m, ["slot_mu", "slot_log_sigma", "project_q.weight", "project_k.weight",
                            "gru.weight_ih", "gru.weight_hh", "gru.bias_hh", "mlp.0.weight",
                            "predictor.blocks.0.attn.proj_q.weight", "predictor.layer_norm.weight"]))

    # 7. whole Motionformer (video_model_builder.py:1103-1353), reduced: D=64, depth 3, ORViT at 1,
    #    crop 64 (exercises the bicubic pos-embed path), 4 input frames
    load_motionformer(mods)
    cfg = small_cfg()
    torch.manual_seed(0)
    m = mods["video_model_builder"].Motionformer(cfg)
    randomize(m, g, std=0.1)
    m.train()
    x = torch.randn(2, 3, 4, 64, 64, generator=g)
    boxes = make_boxes(g, 2, 4, 3)
    labels = torch.tensor([3, 7])
    logits = m([x], {"orvit_bboxes": boxes.clone()})
    loss = mods_loss(logits, labels)
    loss.backward()
    keys = ["patch_embed_3d.proj.weight", "pos_embed", "temp_embed", "cls_token", "blocks.0.attn.qkv.weight",
            "blocks.1.patch_to_d.2.weight", "blocks.1.motion_stream.c_coord_to_feature.0.weight",
            "blocks.2.mlp.fc2.weight", "head.weight", "pre_logits.fc.bias"]
    np.savez(os.path.join(OUT, "motionformer_small.npz"), x=x.numpy(), boxes=boxes.numpy(),
             labels=labels.numpy(), logits=logits.detach().numpy(), loss=loss.detach().numpy(),
             **pack("p.", m.state_dict()), **grads_of(m, keys))
    names = np.array(sorted(m.state_dict().keys()))
    shapes = np.array([",".join(map(str, m.state_dict()[k].shape)) for k in names])
    np.savez(os.path.join(OUT, "motionformer_small_keys.npz"), names=names, shapes=shapes)

    # 8. checkpoint ABI of the full-size model (names + shapes only; SSv2_ORViT-MF_224_16x4.yaml)
    cfg = small_cfg(crop=224, O=4, T=8)
    cfg.ORVIT.LAYERS = [1, 6, 10]
    cfg.MF.EMBED_DIM, cfg.MF.DEPTH, cfg.MF.NUM_HEADS, cfg.MODEL.NUM_CLASSES = 768, 12, 12, 174
    cfg.MF.DROP_PATH = 0.2
    m = mods["video_model_builder"].Motionformer(cfg)
    sd = m.state_dict()
    names = np.array(sorted(sd.keys()))
    shapes = np.array([",".join(map(str, sd[k].shape)) for k in names])
    np.savez(os.path.join(OUT, "motionformer_224_keys.npz"), names=names, shapes=shapes,
             nparams=sum(v.numel() for v in m.parameters()))
    print("wrote", sorted(os.listdir(OUT)))


def main_r2():
    """Round-2 fixtures (own generator, so the round-1 files above regenerate bit-identically)."""
    torch.set_grad_enabled(True)
    mods = load_reference(_roi_align_tv)
    att = mods["attention"]
    g = torch.Generator().manual_seed(20262)

    # 9. trajectory attention with MORE than 224 keys per frame (the HR regime: key tiles + online softmax in the
    #    fused kernels), two heads of 64: thw = [2, 230, 1]  (attention.py:479-557)
    m = att.TrajectoryAttention(128, num_heads=2, qkv_bias=True).double()
    randomize(m, g, std=0.12)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(prm.float().double())                       # fp32-representable: stored as float32
    x = torch.randn(1, 1 + 2 * 230, 128, generator=g).double().requires_grad_(True)
    ct = torch.randn(1, 461, 128, generator=g).double()
    y, _ = m(x, [2, 230, 1])
    (y * ct).sum().backward()
    np.savez(os.path.join(OUT, "traj_attn_p230.npz"), x=x.detach().numpy().astype(np.float32),
             ct=ct.numpy().astype(np.float32), y=y.detach().numpy(), dx=x.grad.numpy(), thw=np.array([2, 230, 1]),
             heads=2, **{k: v.astype(np.float32) for k, v in pack("p.", m.state_dict()).items()},
             **grads_of(m, ["qkv.bias", "proj_q.weight", "proj_kv.weight"]))

    # 10. HR-like Motionformer (EK_ORVIT_MF_HR.yaml reduced): crop 256 != 224 (bicubic pos-embed, 16x16 = 256 > 224
    #     patches per frame), O=6 objects, EPIC-Kitchens verb/noun heads + EKLoss (losses.py:62-95, the reference's own
    #     losses.py is executed), one head of 64 channels, depth 2 with ORViT at layer 1
    load_motionformer(mods)
    losses = _load_losses()
    cfg = small_cfg(crop=256, O=6, T=2)
    cfg.MF.EMBED_DIM, cfg.MF.NUM_HEADS, cfg.MF.DEPTH = 64, 1, 2
    cfg.TRAIN.DATASET = "epickitchens"
    cfg.MODEL.NUM_CLASSES = 97
    cfg.MODEL.LOSS_FUNC = "label_smoothing_cross_entropy"
    cfg.MIXUP = ns(LABEL_SMOOTH_VALUE=0.1)
    torch.manual_seed(0)
    m = mods["video_model_builder"].Motionformer(cfg)
    randomize(m, g, std=0.1)
    m.train()
    x = torch.randn(1, 3, 4, 256, 256, generator=g).half().float()      # stored exactly as fp16
    boxes = make_boxes(g, 1, 4, 6, zero=((0, None, 5),))
    labels = {"verb": torch.tensor([41]), "noun": torch.tensor([207])}
    preds, extra = m([x], {"orvit_bboxes": boxes.clone()})
    loss_fun = losses.get_loss_func(cfg)(reduction="mean")
    ld = loss_fun(extra, labels)
    loss = ld["verb_loss"] + ld["noun_loss"]                              # train_net.py:95-97
    loss.backward()
    keys = ["patch_embed_3d.proj.weight", "pos_embed", "temp_embed", "blocks.0.attn.qkv.weight",
            "blocks.0.attn.proj_kv.weight", "blocks.1.attn.qkv.weight", "blocks.1.patch_to_d.0.weight",
            "blocks.1.box_categories", "blocks.1.motion_mlp.fc2.weight", "head0.weight", "head1.weight",
            "head1.bias", "pre_logits.fc.weight"]
    np.savez(os.path.join(OUT, "motionformer_hr_small.npz"), x=x.numpy().astype(np.float16), boxes=boxes.numpy(),
             label_verb=labels["verb"].numpy(), label_noun=labels["noun"].numpy(),
             verb=extra["verb"].detach().numpy(), noun=extra["noun"].detach().numpy(),
             verb_loss=ld["verb_loss"].detach().numpy(), noun_loss=ld["noun_loss"].detach().numpy(),
             loss=loss.detach().numpy(), **pack("p.", m.state_dict()), **grads_of(m, keys))
    print("wrote round-2 fixtures")



def steve_cfg():
    """Reduced STEVE config: 16 px frames -> 4x4 = 16 dVAE tokens, 8x8 = 64 CNN cells, vocabulary 32, width 32."""
    return ns(SLOTS=ns(NUM_ITERS=2, NUM_SLOTS=3, CNN_HID_SIZE=16, SIZE=16, DIM=32, MLP_HID_SIZE=32, IMG_CHANNELS=3,
                       IMG_SIZE=16, VOCAB_SIZE=32, NUM_PREDICTOR_BLOCKS=1, NUM_PREDICTOR_HEADS=2, PREDICTOR_DROPOUT=0.0,
                       DECODER=ns(DIM=32, NUM_BLOCKS=2, NUM_HEADS=2, DROPOUT=0.1)),
              MODEL=ns(CNN_NAME="base"), TRAIN=ns(MIXED_PRECISION=False))


def main_steve():
    """11. STEVE.forward (steve.py:253-330) of the reference's registered STEVE class at a reduced shape, eval mode (the
    decoder / positional dropouts draw nothing), fp64 with fp32-representable weights; the three random draws forward()
    makes -- two Exp(1) tensors in gumbel_softmax (utils.py:51) and the N(0,1) slot initialisation (steve.py:56) -- are
    captured by drawing them first from the same seed in the same order.  Loss = mse + cross_entropy as in
    tools/steve_train_net.py:97-103."""
    torch.set_grad_enabled(True)
    mods = load_reference(_roi_align_tv)
    g = torch.Generator().manual_seed(20263)
    cfg = steve_cfg()
    torch.manual_seed(3)
    m = mods["steve"].STEVE(cfg).double()
    with torch.no_grad():
        for n, prm in m.named_parameters():
            if not prm.requires_grad:
                continue
            if prm.dim() == 1 and ("norm" in n) and n.endswith("weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=g))
            elif n.endswith("bias"):
                prm.copy_(0.05 * torch.randn(prm.shape, generator=g))
            else:
                prm.copy_((prm + 0.02 * torch.randn(prm.shape, generator=g)).float().double())
            prm.copy_(prm.float().double())
    m.eval()
    B, T, C, S = 2, 2, 3, 16
    video = torch.rand(B, T, C, S, S, generator=g).double()
    tau, hard = 0.7, True
    torch.manual_seed(91)
    e1 = torch.empty(B * T, 32, S // 4, S // 4, dtype=torch.float64).exponential_()
    e2 = torch.empty(B * T, 32, S // 4, S // 4, dtype=torch.float64).exponential_()
    n0 = torch.empty(B, 3, 16, dtype=torch.float64).normal_()
    torch.manual_seed(91)
    recon, ce, mse, attns = m(video, tau, hard)
    (mse + ce).backward()
    keys = ["dvae.encoder.0.m.weight", "dvae.encoder.7.bias", "dvae.decoder.1.m.weight", "dvae.decoder.11.weight",
            "steve_encoder.cnn.fenc.0.m.weight", "steve_encoder.cnn.fenc.3.bias", "steve_encoder.pos.projection.weight",
            "steve_encoder.layer_norm.weight", "steve_encoder.mlp.0.weight", "steve_encoder.savi.project_k.weight",
            "steve_encoder.savi.gru.weight_hh", "steve_encoder.savi.slot_mu", "steve_encoder.slot_proj.weight",
            "steve_decoder.dict.dictionary.weight", "steve_decoder.bos", "steve_decoder.pos.pe",
            "steve_decoder.tf.blocks.0.self_attn.proj_q.weight", "steve_decoder.tf.blocks.1.self_attn.proj_k.weight",
            "steve_decoder.tf.blocks.1.encoder_decoder_attn.proj_k.weight",
            "steve_decoder.tf.blocks.1.encoder_decoder_attn_layer_norm.bias", "steve_decoder.tf.blocks.1.ffn.0.weight",
            "steve_decoder.tf.layer_norm.weight", "steve_decoder.head.weight"]
    sd = {k: v for k, v in m.state_dict().items() if v.is_floating_point()}
    np.savez(os.path.join(OUT, "steve_forward_small.npz"), video=video.numpy().astype(np.float32), tau=tau, hard=hard,
             gumbel_soft=e1.numpy(), gumbel_hard=e2.numpy(), slots_noise=n0.numpy(), recon=recon.detach().numpy(),
             cross_entropy=ce.detach().numpy(), mse=mse.detach().numpy(), attns=attns.detach().numpy(),
             state_keys=np.array(sorted(m.state_dict().keys())),
             **{k: v.astype(np.float32) for k, v in pack("p.", sd).items()}, **grads_of(m, keys))
    print("wrote steve_forward_small.npz")



def main_data():
    """12. Data-side contract: the reference's own datasets/utils.spatial_sampling (train branch and
    the three test views, with boxes), pack_pathway_output and the ssv2.py:337-346 box hand-off (box_ops), executed from
    the reference files with numpy's global RNG seeded; frames 5x3x40x56 float32, boxes [5,3,4] xyxy pixels."""
    from oracle._ref_loader import _load, _ns
    mods = load_reference(_roi_align_tv)
    for n in ("slowfast.datasets", "torchvision.transforms", "torchvision.transforms.functional", "cv2", "slowfast.utils.env"):
        _ns(n)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["slowfast.utils.env"].pathmgr = None
    for stub in ("rand_augment", "boxes_autoaugment", "random_erasing"):
        m = _ns("slowfast.datasets." + stub)
        m.rand_augment_transform = lambda *a, **k: None
        m.RandomErasing = object
    _load("slowfast.datasets.transform", "slowfast/datasets/transform.py")
    du = _load("slowfast.datasets.utils", "slowfast/datasets/utils.py")
    bo = mods["box_ops"]
    g = torch.Generator().manual_seed(20264)
    T, O, H, W = 5, 3, 40, 56
    frames = torch.rand(T, 3, H, W, generator=g)
    x0 = torch.rand(T, O, 1, generator=g) * 30
    y0 = torch.rand(T, O, 1, generator=g) * 20
    wh = torch.rand(T, O, 2, generator=g) * 24 + 0.5
    boxes = torch.cat([x0, y0, x0 + wh[..., :1], y0 + wh[..., 1:]], -1).numpy().astype(np.float32)
    boxes[1, 2] = 0                                                     # an absent object
    boxes[3, 0] = [10.0, 5.0, 10.8, 30.0]                               # thinner than eps after normalisation
    out = {"frames": frames.numpy(), "boxes": boxes}
    cfg = ns(DATA=ns(REVERSE_INPUT_CHANNEL=True), MODEL=ns(ARCH="mformer", SINGLE_PATHWAY_ARCH=["mformer"], MULTI_PATHWAY_ARCH=["slowfast"]))
    for tag, seed, kw in (("train", 7, dict(spatial_idx=-1, min_scale=36, max_scale=48, crop_size=32)),
                          ("train_inv", 8, dict(spatial_idx=-1, min_scale=36, max_scale=48, crop_size=32, inverse_uniform_sampling=True)),
                          ("test0", 9, dict(spatial_idx=0, min_scale=36, max_scale=36, crop_size=32)),
                          ("test1", 9, dict(spatial_idx=1, min_scale=36, max_scale=36, crop_size=32)),
                          ("test2", 9, dict(spatial_idx=2, min_scale=36, max_scale=36, crop_size=32))):
        np.random.seed(seed)
        f, b = du.spatial_sampling(frames.clone(), boxes=boxes.copy(), random_horizontal_flip=True, **kw)
        packed = du.pack_pathway_output(cfg, f.permute(1, 0, 2, 3))[0]   # C T H W as the datasets hand it over
        h, w = packed.shape[-2:]
        bb = b.copy()
        bb[..., [0, 2]] = bb[..., [0, 2]] / w                            # ssv2.py:337-346
        bb[..., [1, 3]] = bb[..., [1, 3]] / h
        bb = np.clip(bb, 0, 1)
        ob = bo.zero_empty_boxes(bo.box_xyxy_to_cxcywh(torch.from_numpy(bb)), mode="cxcywh")
        out.update({tag + ".frames": packed.numpy(), tag + ".boxes_px": b, tag + ".orvit_bboxes": ob.numpy(), tag + ".seed": seed})
    out["norm"] = du.tensor_normalize((frames * 255).to(torch.uint8).permute(0, 2, 3, 1), [0.45, 0.45, 0.45], [0.225, 0.225, 0.225]).numpy()
    np.savez(os.path.join(OUT, "data_contract.npz"), **out)
    print("wrote data_contract.npz")


def main_metrics():
    """13. Evaluation metrics: the reference's own slowfast/utils/metrics.py (scipy comb, per-clip numpy
    tables) on seeded masks and scores -- FG-ARI incl. a perfect and a single-cluster case, top-k / multitask top-k counts."""
    from oracle._ref_loader import _load
    load_reference(_roi_align_tv)
    mt = _load("slowfast.utils.metrics", "slowfast/utils/metrics.py")
    g = torch.Generator().manual_seed(20265)
    B, N0, N1, D = 5, 6, 7, 4 * 16 * 16
    seg = torch.randint(0, N0, (B, D), generator=g)
    true = torch.nn.functional.one_hot(seg, N0).permute(0, 2, 1).float()            # [B,N0,D]
    pred = torch.rand(B, N1, D, generator=g)
    pred[0] = 0.0
    pred[0, :N0] = true[0] * 5.0                                                      # clip 0: perfect prediction
    pred[1] = 0.0
    pred[1, 3] = 1.0                                                                  # clip 1: everything in one slot
    pred[2, :N0] += true[2] * 0.6                                                     # clip 2: partially right
    aris = [mt.evaluate_ari(true[b:b + 1, 1:], pred[b:b + 1]) for b in range(B)]      # foreground segments only
    scores = torch.randn(64, 23, generator=g)
    labels = torch.randint(0, 23, (64,), generator=g)
    s2 = torch.randn(64, 11, generator=g)
    l2 = torch.randint(0, 11, (64,), generator=g)
    np.savez(os.path.join(OUT, "metrics.npz"), true=true.numpy().astype(np.uint8), pred=pred.numpy(),
             ari_each=np.array(aris), ari_fg=mt.evaluate_ari(true[:, 1:], pred), ari_all=mt.evaluate_ari(true, pred),
             scores=scores.numpy(), labels=labels.numpy(), scores2=s2.numpy(), labels2=l2.numpy(),
             topk=np.array([float(x) for x in mt.topks_correct(scores, labels, (1, 5))]),
             topk_acc=np.array([float(x) for x in mt.topk_accuracies(scores, labels, (1, 5))]),
             multitask=np.array([float(x) for x in mt.multitask_topks_correct((scores, s2), (labels, l2), (1, 5))]))
    print("wrote metrics.npz")

def main_ckpt():
    """14. Checkpoint I/O: the reference's own slowfast/utils/checkpoint.py -- save_checkpoint (:112-159)
    WRITES tests/golden/ckpt_small.pyth from its Motionformer carrying the motionformer_small fixture weights and a stepped
    AdamW; its load_checkpoint (:201-394, split_qkv :586-597) then loads (a) that file for a resume and (b) a "pre-trained"
    variant with a `module.` prefix, a 174-way head and the epoch_reset / clear_name_pattern / replace_name_pattern /
    load_orvit_attn_from_bb switches; the resulting model states are the expected outputs (ckpt_small_expected.npz).
    pathmgr (iopath) is plumbing: a stand-in over os / open."""
    from oracle._ref_loader import _load, _ns
    mods = load_motionformer(load_reference(_roi_align_tv))
    env = _ns("slowfast.utils.env")

    class _PM:
        exists = staticmethod(os.path.exists)
        mkdirs = staticmethod(lambda p: os.makedirs(p, exist_ok=True))
        ls = staticmethod(os.listdir)
        open = staticmethod(open)
    env.checkpoint_pathmgr = _PM
    env.pathmgr = _PM
    _load("slowfast.utils.c2_model_loading", "slowfast/utils/c2_model_loading.py")
    ck = _load("slowfast.utils.checkpoint", "slowfast/utils/checkpoint.py")
    z = np.load(os.path.join(OUT, "motionformer_small.npz"))
    params = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    cfg = small_cfg()
    cfg.NUM_GPUS, cfg.NUM_SHARDS = 1, 1
    cfg.dump = lambda: "MODEL:\n  MODEL_NAME: Motionformer\n"
    torch.manual_seed(0)
    m = mods["video_model_builder"].Motionformer(cfg)
    m.load_state_dict(params)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(4)
    for p_ in m.parameters():
        p_.grad = torch.randn(p_.shape, generator=g) * 0.01
    opt.step()                                            # a real optimizer state (step, exp_avg, exp_avg_sq) ...
    m.load_state_dict(params)                             # ... around the fixture's weights
    import tempfile
    tmp = tempfile.mkdtemp()
    path = ck.save_checkpoint(tmp, m, opt, 3, cfg, name="checkpoint_epoch_00004")
    import shutil
    shutil.copy(path, os.path.join(OUT, "ckpt_small.pyth"))
    # (a) resume into a differently initialised model
    torch.manual_seed(1)
    m2 = mods["video_model_builder"].Motionformer(cfg)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    epoch = ck.load_checkpoint(path, m2, data_parallel=False, optimizer=opt2)
    out = {"resume_epoch": np.int64(epoch)}
    out.update(pack("resume.", m2.state_dict()))
    out["resume_opt_step"] = np.float64(float(opt2.state_dict()["state"][0]["step"]))
    out["resume_opt_exp_avg0"] = opt2.state_dict()["state"][0]["exp_avg"].numpy()
    # (b) "pre-trained backbone" file: DDP prefix, a 174-way head, backbone attention to be copied into orvit_ names
    sd = OrderedDict(("module." + k, v.clone()) for k, v in m.state_dict().items())
    sd["module.head.weight"] = torch.zeros(174, 64)
    sd["module.head.bias"] = torch.zeros(174)
    sd["module.unknown.weight"] = torch.ones(3)
    pre = os.path.join(tmp, "pretrained.pyth")
    torch.save({"epoch": 9, "model_state": sd, "optimizer_state": {}, "cfg": ""}, pre)
    shutil.copy(pre, os.path.join(OUT, "ckpt_small_pretrained.pyth"))
    torch.manual_seed(2)
    m3 = mods["video_model_builder"].Motionformer(cfg)
    before = {k: v.clone() for k, v in m3.state_dict().items()}
    epoch3 = ck.load_checkpoint(pre, m3, data_parallel=False, epoch_reset=True, clear_name_pattern=("module.",),
                                replace_name_pattern=(("nonexistent_a", "nonexistent_b"),), load_orvit_attn_from_bb=True)
    out["finetune_epoch"] = np.int64(epoch3)
    out.update(pack("finetune.", m3.state_dict()))
    out["finetune_untouched"] = np.array(sorted(k for k, v in m3.state_dict().items() if torch.equal(v, before[k])))
    # split_qkv on its own (:586-597)
    sq = ck.split_qkv(OrderedDict([("blocks.0.attn.qkv.weight", torch.arange(24.).reshape(6, 4)), ("x", torch.ones(2))]))
    out["split_keys"] = np.array(list(sq.keys()))
    out["split_q"] = sq["blocks.0.attn.q.weight"].numpy()
    out["split_v"] = sq["blocks.0.attn.v.weight"].numpy()
    np.savez(os.path.join(OUT, "ckpt_small_expected.npz"), **out)
    print("wrote ckpt_small.pyth, ckpt_small_pretrained.pyth, ckpt_small_expected.npz")


def main_mvit():
    """MViT with ORViT blocks (video_model_builder.py:765-1101) at a reduced size: the reference's own MViT, MultiScaleBlock /
    MultiScaleAttention (attention.py:16-352), round_width (utils.py:31-44, the real function: its module's two config-file
    imports get empty stand-ins) and ORViT, one ORViT in place of a block and one beside a block."""
    torch.set_grad_enabled(True)
    mods = load_reference(_roi_align_tv)
    load_motionformer(mods)
    import sys as _sys
    from oracle._ref_loader import _load, _ns
    ed = _ns("easydict")
    ed.EasyDict = dict
    real_utils = _load("slowfast.models.utils", "slowfast/models/utils.py")
    vmb = mods["video_model_builder"]
    vmb.round_width = real_utils.round_width
    misc = _ns("slowfast.utils.misc")
    misc.get_num_classes = lambda cfg: cfg.MODEL.NUM_CLASSES          # misc.py:417-421, non-EPIC branch
    _sys.modules["slowfast.utils"].misc = misc
    # As shipped the reference's MViT cannot be constructed: MultiScaleBlock passes drop_rate= to common.Mlp, whose keyword
    # is drop= (attention.py:321-327 vs common.py:8-15: TypeError).  The keyword is mapped here; the arithmetic is the file's.
    att = mods["attention"]
    real_mlp = att.Mlp
    att.Mlp = lambda *a, **k: real_mlp(*a, **{('drop' if n == 'drop_rate' else n): v for n, v in k.items()})
    g = torch.Generator().manual_seed(20263)
    cfg = small_cfg(crop=64, O=3, T=2)
    cfg.DATA.TEST_CROP_SIZE = 64
    cfg.DATA.INPUT_CHANNEL_NUM = [3]
    cfg.ORVIT.LAYERS, cfg.ORVIT.ADD_LAYERS = [3], [2]
    cfg.DETECTION = ns(ENABLE=False)
    cfg.MODEL.DROPOUT_RATE, cfg.MODEL.HEAD_ACT = 0.0, "softmax"
    cfg.MVIT = ns(MODE="conv", POOL_FIRST=False, CLS_EMBED_ON=True, PATCH_KERNEL=[3, 7, 7], PATCH_STRIDE=[2, 4, 4],
                  PATCH_PADDING=[1, 3, 3], PATCH_2D=False, EMBED_DIM=32, NUM_HEADS=2, MLP_RATIO=4.0, QKV_BIAS=True,
                  DROPPATH_RATE=0.0, DEPTH=4, NORM="layernorm", DIM_MUL=[[1, 2.0]], HEAD_MUL=[[1, 2.0]], POOL_KV_STRIDE=None,
                  POOL_KV_STRIDE_ADAPTIVE=[1, 4, 4], POOL_Q_STRIDE=[[1, 1, 2, 2]], POOL_KVQ_KERNEL=[3, 3, 3],
                  ZERO_DECAY_POS_CLS=False, NORM_STEM=False, SEP_POS_EMBED=True, DROPOUT_RATE=0.0,
                  POOL_KV_IGNORE_111_KERNEL=False)
    torch.manual_seed(0)
    m = vmb.MViT(cfg)
    randomize(m, g, std=0.1)
    m.train()
    x = torch.randn(2, 3, 4, 64, 64, generator=g).half().float()
    boxes = make_boxes(g, 2, 4, 3)
    ct = torch.randn(2, 10, generator=g)
    y = m([x], {"orvit_bboxes": boxes.clone()})
    (y * ct).sum().backward()
    keys = ["patch_embed.proj.weight", "pos_embed_spatial", "pos_embed_temporal", "cls_token", "blocks.0.attn.q.weight",
            "blocks.0.attn.pool_k.weight", "blocks.0.attn.norm_k.weight", "blocks.0.proj.weight", "blocks.1.attn.pool_q.weight",
            "blocks.1.attn.norm_q.bias", "blocks.1.mlp.fc1.weight", "blocks.2.attn.v.bias", "orvit_blocks.2.attn.qkv.weight",
            "orvit_blocks.2.box_categories", "blocks.3.attn.proj_kv.weight", "blocks.3.patch_to_d.0.weight",
            "head.projection.weight"]
    np.savez(os.path.join(OUT, "mvit_orvit_small.npz"), x=x.numpy().astype(np.float16), boxes=boxes.numpy(), ct=ct.numpy(),
             y=y.detach().numpy(), **pack("p.", m.state_dict()), **grads_of(m, keys))
    print("wrote mvit_orvit_small.npz", tuple(y.shape))


def _load_losses():
    """The reference's own slowfast/models/losses.py (plain torch + the stubbed logger)."""
    from oracle._ref_loader import _load
    return _load("slowfast.models.losses", "slowfast/models/losses.py")


def mods_loss(logits, labels):
    """LabelSmoothingCrossEntropy (losses.py:40-59) is plain torch; restated inline to avoid loading
    losses.py's unrelated imports."""
    lp = torch.log_softmax(logits, dim=-1)
    nll = -lp.gather(-1, labels.unsqueeze(1)).squeeze(1)
    return (0.9 * nll + 0.1 * (-lp.mean(-1))).mean()


if __name__ == "__main__":
    if "--mvit-only" in sys.argv:
        main_mvit()
        sys.exit(0)
    if "--ckpt-only" in sys.argv:
        main_ckpt()
    elif "--metrics-only" in sys.argv:
        main_metrics()
    elif "--data-only" in sys.argv:
        main_data()
    elif "--steve-only" in sys.argv:
        main_steve()
    else:
        if "--r2-only" not in sys.argv:
            main()
        main_r2()
        main_steve()
        main_data()
        main_metrics()
        main_ckpt()
