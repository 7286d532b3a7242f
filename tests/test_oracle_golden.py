"""Pins the CPU oracle (oracle/focus_oracle.py) to fixtures produced by the reference's own modules
(oracle/make_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import load_golden


def T(a, dtype=None, grad=False):
    t = torch.from_numpy(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(grad)


def close(a, b, tol):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-3)  # floor: some grads are analytically 0
    assert err < tol, err


def leafify(p):
    return {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in p.items()}


def check_grads(arrs, p, tol):
    for k, g in arrs.items():
        if k.startswith("grad."):
            close(p[k[5:]].grad, g, tol)


def test_trajectory_attention(oracle):
    for fx in ("traj_attn", "traj_attn_p21"):
        a, p = load_golden(fx)
        p = leafify(p)
        x = T(a["x"], grad=True)
        y = oracle.trajectory_attention(p, "", x, list(a["thw"]), int(a["heads"])) if False else \
            oracle.trajectory_attention({("." + k): v for k, v in p.items()}, "", x, list(a["thw"]), int(a["heads"]))
        close(y, a["y"], 1e-12)
        (y * T(a["ct"])).sum().backward()
        close(x.grad, a["dx"], 1e-11)
        check_grads(a, p, 1e-11)
        # the v2 half of proj_kv is dead under use_original_code=True (gradient exactly zero)
        C = x.shape[-1]
        assert float(p["proj_kv.weight"].grad[C:].abs().max()) == 0.0


def test_trajectory_block(oracle):
    a, p = load_golden("traj_block")
    p = leafify(p)
    x = T(a["x"], grad=True)
    y = oracle.trajectory_block({("b." + k): v for k, v in p.items()}, "b", x, list(a["thw"]), int(a["heads"]))
    close(y, a["y"], 1e-12)
    (y * T(a["ct"])).sum().backward()
    close(x.grad, a["dx"], 1e-11)
    check_grads(a, p, 1e-11)


def test_joint_block(oracle):
    a, p = load_golden("joint_block")
    p = leafify(p)
    x = T(a["x"], grad=True)
    y = oracle.joint_attention_block({("b." + k): v for k, v in p.items()}, "b", x, int(a["heads"]))
    close(y, a["y"], 1e-12)
    (y * T(a["ct"])).sum().backward()
    close(x.grad, a["dx"], 1e-11)
    check_grads(a, p, 1e-11)


def test_box_layout_and_closed_form(oracle):
    a, _ = load_golden("box_layout")
    vecs = T(a["vecs"], grad=True)
    boxes = T(a["boxes"])
    B, C, Tn, H, W = a["out"].shape
    out = oracle.box_layout(vecs, boxes, H, W)                     # [B,T,H,W,C]
    close(out.permute(0, 4, 1, 2, 3), a["out"], 1e-6)
    (out.permute(0, 4, 1, 2, 3) * T(a["ct"])).sum().backward()
    close(vecs.grad, a["dvecs"], 1e-6)
    wy, wx, keep = oracle.box_layout_weights(boxes, H, W)
    cf = torch.einsum("btoc,btoy,btox->btyxc", vecs.detach() * keep[..., None], wy, wx)
    close(cf.permute(0, 4, 1, 2, 3), a["out"], 2e-6)


def test_orvit_block(oracle):
    a, p = load_golden("orvit_block")
    p = leafify(p)
    x = T(a["x"], grad=True)
    y = oracle.orvit_block({("b." + k): v for k, v in p.items()}, "b", x, T(a["boxes"]), list(a["thw"]),
                           int(a["heads"]), int(a["crop"]))
    close(y, a["y"], 2e-6)
    (y * T(a["ct"])).sum().backward()
    close(x.grad, a["dx"], 2e-5)
    check_grads(a, p, 2e-5)


def test_slot_attention_video(oracle):
    a, p = load_golden("slot_attention")
    p = leafify(p)
    inp = T(a["inputs"], grad=True)
    slots, attns = oracle.slot_attention_video(p, inp, T(a["noise"]), int(a["iters"]), int(a["pred_heads"]),
                                               int(a["pred_blocks"]))
    close(slots, a["slots"], 1e-12)
    close(attns, a["attns"], 1e-12)
    ((slots * T(a["ct_slots"])).sum() + (attns * T(a["ct_attns"])).sum()).backward()
    close(inp.grad, a["dinputs"], 1e-10)
    check_grads(a, p, 1e-10)


def test_motionformer_small(oracle):
    a, p = load_golden("motionformer_small")
    p = leafify(p)
    cfg = dict(depth=3, heads=4, orvit_layers=[1], temporal_resolution=2, patch=(2, 16, 16), crop=64)
    logits = oracle.motionformer_forward(p, T(a["x"]), T(a["boxes"]), cfg, training=True)
    close(logits, a["logits"], 5e-6)
    loss = oracle.label_smoothing_ce(logits, T(a["labels"]))
    close(loss, a["loss"], 1e-6)
    loss.backward()
    check_grads(a, p, 1e-4)


def test_trajectory_attention_more_than_224_keys(oracle):
    """HR regime (P = 230 > 224 keys per frame, two heads of 64): fixture from the reference module in fp64."""
    a, p = load_golden("traj_attn_p230", dtype=torch.float64)
    p = leafify(p)
    x = T(a["x"], dtype=torch.float64, grad=True)
    y = oracle.trajectory_attention({("." + k): v for k, v in p.items()}, "", x, list(a["thw"]), int(a["heads"]))
    close(y, a["y"], 1e-12)
    (y * T(a["ct"], dtype=torch.float64)).sum().backward()
    close(x.grad, a["dx"], 1e-11)
    check_grads(a, p, 1e-11)


def test_motionformer_hr_small_ek_heads(oracle):
    """Reduced EK_ORVIT_MF_HR: crop 256 (bicubic pos-embed, 256 patches + 6 objects per frame), verb/noun heads,
    EKLoss -- logits, both losses, their sum and selected gradients from the reference model + its own losses.py."""
    a, p = load_golden("motionformer_hr_small")
    p = leafify(p)
    cfg = dict(depth=2, heads=1, orvit_layers=[1], temporal_resolution=2, patch=(2, 16, 16), crop=256)
    verb, extra = oracle.motionformer_forward(p, T(a["x"], dtype=torch.float32), T(a["boxes"]), cfg, training=True)
    close(extra["verb"], a["verb"], 2e-5)
    close(extra["noun"], a["noun"], 2e-5)
    ld = oracle.ek_loss(extra, {"verb": T(a["label_verb"]), "noun": T(a["label_noun"])})
    close(ld["verb_loss"], a["verb_loss"], 1e-6)
    close(ld["noun_loss"], a["noun_loss"], 1e-6)
    loss = ld["verb_loss"] + ld["noun_loss"]
    close(loss, a["loss"], 1e-6)
    loss.backward()
    check_grads(a, p, 2e-4)


STEVE_CFG = dict(img_size=16, num_slots=3, num_iters=2, pred_heads=2, pred_blocks=1, dec_heads=2, dec_blocks=2)


def steve_noise(a, dtype=torch.float64):
    return {"gumbel_soft": T(a["gumbel_soft"], dtype), "gumbel_hard": T(a["gumbel_hard"], dtype),
            "slots": T(a["slots_noise"], dtype)}


def test_steve_forward(oracle):
    """STEVE.forward (steve.py:253-330): fixture from the reference's registered STEVE class, eval-mode dropout, the two
    Gumbel draws and the slot initialisation captured; recon, cross entropy, mse, attns and 23 parameter gradients."""
    a, p = load_golden("steve_forward_small", dtype=torch.float64)
    p = leafify(p)
    recon, ce, mse, attns = oracle.steve_forward(p, T(a["video"], torch.float64), float(a["tau"]), bool(a["hard"]),
                                                 steve_noise(a), STEVE_CFG)
    close(recon, a["recon"], 1e-12)
    close(attns, a["attns"], 1e-12)
    close(ce, a["cross_entropy"], 1e-12)
    close(mse, a["mse"], 1e-12)
    (mse + ce).backward()
    check_grads(a, p, 1e-10)


def test_steve_state_dict_keys_match_the_reference():
    """The registered STEVE class builds the reference's module tree: same state_dict keys and shapes (checkpoints of the
    reference load unchanged).  Construction needs no GPU."""
    from focus_amd.slowfast.config.defaults import get_cfg
    from focus_amd.slowfast.models import MODEL_REGISTRY
    a, p = load_golden("steve_forward_small")
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "STEVE"
    s = cfg.SLOTS
    s.NUM_ITERS, s.NUM_SLOTS, s.CNN_HID_SIZE, s.SIZE, s.DIM, s.MLP_HID_SIZE, s.IMG_SIZE, s.VOCAB_SIZE = 2, 3, 16, 16, 32, 32, 16, 32
    s.NUM_PREDICTOR_BLOCKS, s.NUM_PREDICTOR_HEADS = 1, 2
    s.DECODER.DIM, s.DECODER.NUM_BLOCKS, s.DECODER.NUM_HEADS = 32, 2, 2
    m = MODEL_REGISTRY.get("STEVE")(cfg)
    sd = m.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in a["state_keys"]]
    for k, v in p.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k


MVIT_SMALL = dict(embed_dim=32, heads=2, depth=4, dim_mul=[[1, 2.0]], head_mul=[[1, 2.0]], pool_q_stride=[[1, 1, 2, 2]],
                  pool_kvq_kernel=[3, 3, 3], pool_kv_stride_adaptive=[1, 4, 4], patch_kernel=[3, 7, 7], patch_stride=[2, 4, 4],
                  patch_padding=[1, 3, 3], crop=64, frames=4, orvit_layers=[3], orvit_add_layers=[2])


def test_mvit_with_orvit_blocks(oracle):
    """oracle.mvit_forward against the reference's own MViT (video_model_builder.py:765-1101) at a reduced size, one ORViT
    in place of a block and one beside a block (oracle/make_golden.py main_mvit)."""
    a, p = load_golden("mvit_orvit_small", dtype=torch.float64)
    p = leafify(p)
    y = oracle.mvit_forward(p, T(a["x"], torch.float64), T(a["boxes"], torch.float64), MVIT_SMALL, training=True)
    close(y, a["y"], 2e-5)                                      # the fixture is the reference in fp32
    (y * T(a["ct"], torch.float64)).sum().backward()
    check_grads(a, p, 2e-4)
    plan = oracle.mvit_plan(MVIT_SMALL)
    assert [(l["dim"], l["dim_out"], l["heads"]) for l in plan] == [(32, 64, 2), (64, 64, 4), (64, 64, 4), (64, 64, 4)]
    assert oracle.round_width(96, 2.0, divisor=2) == 192 and oracle.round_width(1, 2.0) == 2
