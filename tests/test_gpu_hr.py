"""BASELINE configs[4] regime: ORViT-Motionformer-HR 16x336 -- 21x21 = 441 patches (+6 objects = 447) per frame, more
than the 224 keys one register tile of the fused space-attention kernels holds; EPIC-Kitchens verb/noun heads + EKLoss.
The fused kernels cut each frame into key tiles merged by an online softmax (forward) / streamed with the saved
log-sum-exp (backward); these tests pin that path to reference-generated fixtures and to the CPU oracle, and check that
it is the path taken (no S x S logits workspace)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import DTYPES, TOL, T, check_param_grads, close, dev, load_module

pytestmark = pytest.mark.gpu


def _hr_small_cfg(mixed):
    from focus_amd.slowfast.config.defaults import get_cfg
    cfg = get_cfg()
    cfg.merge_from_list(["ORVIT.ENABLE", True, "ORVIT.O", 6, "ORVIT.LAYERS", [1], "DATA.TRAIN_CROP_SIZE", 256,
                         "DATA.NUM_FRAMES", 4, "MF.EMBED_DIM", 64, "MF.DEPTH", 2, "MF.NUM_HEADS", 1,
                         "MF.TEMPORAL_RESOLUTION", 2, "MF.USE_MLP", True, "MODEL.NUM_CLASSES", 97,
                         "MODEL.MODEL_NAME", "Motionformer", "TRAIN.DATASET", "epickitchens", "NUM_GPUS", 1,
                         "TRAIN.MIXED_PRECISION", mixed, "MODEL.LOSS_FUNC", "label_smoothing_cross_entropy"])
    return cfg


@pytest.mark.parametrize("dtype", DTYPES)
def test_trajectory_attention_p230_golden(dtype):
    """P = 230 keys per frame (8 key blocks = 2 tiles of 4), two heads of 64: reference fixture (fp64 module)."""
    from focus_amd.slowfast.models.attention import TrajectoryAttention
    a, p = load_golden("traj_attn_p230")
    C = a["x"].shape[-1]
    m = load_module(TrajectoryAttention(C, num_heads=int(a["heads"]), qkv_bias=True), p)
    x = T(a["x"]).to(dev(), dtype).requires_grad_()
    y, _ = m(x, [int(v) for v in a["thw"]])
    (y.float() * T(a["ct"]).to(dev())).sum().backward()
    tol = TOL[dtype]
    close(y, a["y"], tol, "y")
    close(x.grad, a["dx"], tol * 2, "dx")
    check_param_grads(m, a, tol * 2)


def test_hr_shapes_take_the_fused_kernels():
    """focus_traj_space_workspace_bytes for the HR shapes holds the cls-row scratch only: the S x S logits of the
    unfused path (B*h*S*S*2 bytes = 307 MB per clip and block at S = 3576) are not part of it."""
    from focus_amd import _lib
    L = _lib.lib()
    for P in (441, 447, 225, 448):
        S = 8 * P
        fwd = L.focus_traj_space_workspace_bytes(1, 8, P, 12, 64, _lib.BF16, 0)
        bwd = L.focus_traj_space_workspace_bytes(1, 8, P, 12, 64, _lib.BF16, 1)
        assert fwd < 12 * S * 2 * 64, (P, fwd)                       # a few rows per head, not S x S
        assert bwd < 12 * S * S * 2 // 8, (P, bwd)                   # lse/delta/dxsum scratch, far below one S x S
    # beyond the fused kernels' 14 key blocks the generic path (and its logits workspace) is used
    assert L.focus_traj_space_workspace_bytes(1, 2, 449, 2, 64, _lib.BF16, 0) >= 2 * 898 * 898 * 2
    # fp32 (precision mode) always takes the generic path
    assert L.focus_traj_space_workspace_bytes(1, 8, 441, 12, 64, _lib.F32, 0) >= 12 * 3528 * 3528 * 4


@pytest.mark.parametrize("P,F_", [(225, 2), (257, 3), (300, 2), (352, 2), (441, 2), (447, 3), (448, 2)])
def test_space_attention_key_tilings(oracle, P, F_):
    """Every key tiling the dispatcher can pick for 224 < P <= 448 (8..14 key blocks: 2x4, 3x3, 2x5, 11x1, 2x7 ...),
    ragged last blocks included, bf16 fused kernels against the oracle on bf16-rounded inputs (fp64 arithmetic):
    x~, x_diag, cls row and the input gradient of the space step alone, held to bf16 output rounding."""
    from focus_amd import ops
    g = torch.Generator().manual_seed(1000 + P)
    heads, d, B = 2, 64, 1
    C, S = heads * d, F_ * P
    qkv = (torch.randn(B, 1 + S, 3 * C, generator=g) * 0.9).bfloat16()
    ctx = torch.randn(B, S, F_, C, generator=g).bfloat16()
    ctd = torch.randn(B, S, C, generator=g).bfloat16()
    ctc = torch.randn(B, 1, C, generator=g).bfloat16()
    # oracle of the space step (attention.py:509-535) in fp64 on the same bf16 values
    q64 = qkv.double().requires_grad_()
    q, k, v = (oracle.split_heads(t, heads) for t in q64.split(C, dim=-1))
    scale = d ** -0.5
    cls_ref = oracle.merge_heads(torch.softmax((q[:, :, :1] * scale) @ k.transpose(-1, -2), dim=-1) @ v)
    A = torch.softmax((q[:, :, 1:] @ k[:, :, 1:].transpose(-1, -2)).reshape(B, heads, S, F_, P) * scale, dim=-1)
    xt_ref = torch.einsum("bhsfp,bhfpd->bhsfd", A, v[:, :, 1:].reshape(B, heads, F_, P, d))
    xt_ref = xt_ref.permute(0, 2, 3, 1, 4).reshape(B, S, F_, C)
    xd_ref = xt_ref[:, torch.arange(S), torch.arange(S) // P]
    ((xt_ref * ctx.double()).sum() + (xd_ref * ctd.double()).sum() + (cls_ref * ctc.double()).sum()).backward()
    qg = qkv.to(dev()).requires_grad_()
    xt, xd, cls = ops.traj_space(qg, F_, P, heads)
    ((xt.float() * ctx.to(dev()).float()).sum() + (xd.float() * ctd.to(dev()).float()).sum()
     + (cls.float() * ctc.to(dev()).float()).sum()).backward()
    # outputs: one bf16 rounding of an O(1) value + bf16 probabilities inside P.V
    assert float((xt.float().cpu() - xt_ref.float()).abs().max()) < 2.5e-2 * float(xt_ref.abs().max())
    assert float((xd.float().cpu() - xd_ref.float()).abs().max()) < 2.5e-2 * float(xd_ref.abs().max())
    assert float((cls.float().cpu() - cls_ref.float()).abs().max()) < 2.5e-2 * float(cls_ref.abs().max())
    close(xt, xt_ref, 1e-2, "xt P=%d" % P)
    close(qg.grad, q64.grad, 2e-2, "dqkv P=%d" % P, floor=1e-2 * float(q64.grad.abs().max()))


@pytest.mark.parametrize("mixed", [False, True])
def test_motionformer_hr_small_golden(mixed):
    """Reduced EK_ORVIT_MF_HR (crop 256: bicubic pos-embed, 256 + 6 tokens per frame, verb/noun heads, EKLoss) against
    the fixture produced by the reference model and the reference's own losses.py."""
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    a, p = load_golden("motionformer_hr_small")
    cfg = _hr_small_cfg(mixed)
    m = build_model(cfg)
    m.load_state_dict({k: v.float() for k, v in p.items()})
    m.train()
    preds, extra = m([T(a["x"]).to(dev())], {"orvit_bboxes": T(a["boxes"]).to(dev())})
    tol = 3e-2 if mixed else 1e-3
    assert preds is extra["verb"]
    close(extra["verb"], a["verb"], tol, "verb logits")
    close(extra["noun"], a["noun"], tol, "noun logits")
    labels = {"verb": torch.from_numpy(a["label_verb"]).to(dev()), "noun": torch.from_numpy(a["label_noun"]).to(dev())}
    ld = get_loss_func(cfg)(reduction="mean")(extra, labels)
    for k in ("verb_loss", "noun_loss"):
        assert abs(float(ld[k].detach()) - float(a[k])) < tol * max(1.0, abs(float(a[k]))), k
    loss = ld["verb_loss"] + ld["noun_loss"]
    assert abs(float(loss.detach()) - float(a["loss"])) < tol * max(1.0, abs(float(a["loss"])))
    loss.backward()
    if mixed:
        # the max over RoI cells is discontinuous: a bf16 rounding can move the arg-max cell and re-route the whole
        # patch_to_d gradient of that channel (see test_orvit_block_golden); sanity-bounded only in bf16
        for k in [k for k in a if k.startswith("grad.") and "patch_to_d" in k]:
            close(dict(m.named_parameters())[k[5:]].grad, a.pop(k), 0.3, k, floor=1e-2)
    check_param_grads(m, a, tol * (3 if mixed else 2))
    # eval mode returns probabilities of both heads (video_model_builder.py:1344-1345)
    m.eval()
    with torch.no_grad():
        pv, ex = m([T(a["x"]).to(dev())], {"orvit_bboxes": T(a["boxes"]).to(dev())})
    assert abs(float(ex["noun"].sum()) - 1.0) < 1e-3 and abs(float(pv.sum()) - 1.0) < 1e-3


def test_motionformer_hr_full_size_vs_oracle(oracle):
    """BASELINE configs[4] shape: the whole ORViT-Motionformer-HR 16x336 (21x21 patches, 6 objects, EK heads, reference
    init scheme) on one synthetic clip, bf16 fused path against the CPU oracle: both heads' logits and the EK loss;
    gradients of a spread of parameters against the oracle's autograd."""
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    from focus_amd.train import synthetic_batch
    import bench
    cfg = bench.make_cfg(1, 1, mixed=True, hr=True)
    torch.manual_seed(0)
    m = build_model(cfg)
    m.train()
    with torch.no_grad():      # the reference init leaves the patch-embed conv weight and box_categories at zero
        g = torch.Generator().manual_seed(5)
        m.patch_embed_3d.proj.weight.copy_(0.02 * torch.randn(m.patch_embed_3d.proj.weight.shape, generator=g))
        for blk in m.blocks:
            if hasattr(blk, "box_categories"):
                blk.box_categories.copy_(0.02 * torch.randn(blk.box_categories.shape, generator=g))
    for mod in m.modules():    # stochastic depth off for the comparison
        if mod.__class__.__name__ == "DropPath":
            mod.drop_prob = 0.0
    inputs, labels, meta = synthetic_batch(cfg, 1, "cpu", seed=7)
    assert meta["orvit_bboxes"].shape == (1, 16, 6, 4) and inputs[0].shape == (1, 3, 16, 336, 336)
    names = ["head0.weight", "head1.bias", "pre_logits.fc.weight", "blocks.11.mlp.fc2.weight", "blocks.11.attn.qkv.weight",
             "blocks.10.patch_to_d.2.weight", "blocks.10.attn.proj_kv.weight", "blocks.10.motion_mlp.fc1.weight"]
    # (all in the last two blocks + head: the oracle's autograd then keeps the S x S tensors of 2 blocks only)
    params = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    for k in names:
        params[k].requires_grad_()
    ocfg = dict(depth=12, heads=12, orvit_layers=[1, 6, 10], temporal_resolution=8, patch=(2, 16, 16), crop=336)
    _, ref = oracle.motionformer_forward(params, inputs[0], meta["orvit_bboxes"], ocfg, training=True)
    rl = oracle.ek_loss(ref, labels)
    ref_loss = rl["verb_loss"] + rl["noun_loss"]
    ref_loss.backward()
    d = dev()
    _, got = m([inputs[0].to(d)], {"orvit_bboxes": meta["orvit_bboxes"].to(d)})
    close(got["verb"], ref["verb"], 3e-2, "HR verb logits")
    close(got["noun"], ref["noun"], 3e-2, "HR noun logits")
    ld = get_loss_func(cfg)(reduction="mean")(got, {k: v.to(d) for k, v in labels.items()})
    loss = ld["verb_loss"] + ld["noun_loss"]
    assert abs(float(loss.detach()) - float(ref_loss)) < 3e-2 * float(ref_loss)
    loss.backward()
    named = dict(m.named_parameters())
    for k in names:
        gr = params[k].grad
        close(named[k].grad, gr, 9e-2, "HR grad " + k, floor=1e-2 * float(gr.abs().max()) + 1e-8)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
