"""fp8-weight path (BASELINE configs[4]: "fp8 MFMA weights"; SURVEY 8(d) config 5: OCP e4m3 weights for the Linear GEMMs,
bf16 activations).  The quantiser against oracle/fp8.py code by code; the GEMM (every epilogue) and the dX GEMM against
fp64 on the SAME e4m3-rounded weights at the bf16 kernel limits of tests/test_gpu_kernels.py; Linear / MLP autograd;
requantisation after an optimizer step; the whole 16x336 O=6 clip against the CPU oracle run on e4m3-rounded weights."""
import re

import numpy as np
import pytest
import torch

from test_gpu_kernels import Check, U, bf, dgelu64, gelu64
from test_gpu_parity import close, dev

pytestmark = pytest.mark.gpu

QUANTISED = re.compile(r"blocks\.\d+\.(attn\.(qkv|proj_q|proj)|mlp\.fc[12]|motion_mlp\.fc[12]|patch_to_d\.[02])\.weight$")


def _deq(w):
    """fp64 tensor of the e4m3-rounded weights the product multiplies by (oracle/fp8.py), on w's device."""
    from oracle import fp8
    return torch.from_numpy(fp8.fake_quant(w.detach().float().cpu().numpy()).astype(np.float64)).to(w.device)


@pytest.mark.parametrize("shape,scale", [((768, 3072), 0.02), ((2304, 768), 1.0), ((64, 64), 300.0), ((384, 768), 1e-4)])
def test_fp8_codes_and_scale_match_oracle(shape, scale):
    from focus_amd import ops
    from oracle import fp8
    ops.drop_caches()
    g = torch.Generator().manual_seed(shape[0])
    w = (torch.randn(*shape, generator=g) * scale)
    w[3, 5] = 0.0
    wd = w.to(dev())
    codes, sc = ops.shadow_fp8(wd)
    codesT, _ = ops.shadow_fp8(wd, transposed=True)
    ref_codes, ref_scale = fp8.quantize_per_tensor(w.numpy())
    got = codes.cpu().numpy()
    zero = (ref_codes & 0x7F) == 0                       # +-0 carry no information
    assert np.array_equal(got[~zero], ref_codes[~zero]) and np.all((got[zero] & 0x7F) == 0)
    assert torch.equal(codesT, codes.t().contiguous())
    assert float(sc) == float(ref_scale)
    # all-zero tensor: scale 1, codes 0
    z = torch.zeros(64, 128, device=dev())
    cz, sz = ops.shadow_fp8(z)
    assert float(sz) == 1.0 and int((cz & 0x7F).max()) == 0


HR_SHAPES = [(14116, 768, 768), (14116, 2304, 768), (14116, 3072, 768), (14116, 768, 3072), (3529, 768, 768)]
RAGGED = [(1601, 768, 768), (12808, 1536, 768), (6272, 384, 768), (21168, 768, 384), (1025, 64, 128)]


@pytest.mark.parametrize("shape", HR_SHAPES + RAGGED)
def test_fp8_nt_gemm_epilogues(shape):
    """C = epi(a . Wq^T * scale + bias) [+ residual] with Wq the e4m3 codes: same structure and limits as
    test_gpu_kernels.test_nt_gemm_epilogues, the fp64 reference multiplies the SAME rounded weights."""
    from focus_amd import _lib, ops
    ops.drop_caches()
    M, N, K = shape
    d = dev()
    g = torch.Generator(device=d).manual_seed(M + N + K)
    a = bf(torch.randn(M, K, device=d, generator=g))
    w = torch.randn(N, K, device=d, generator=g) * K ** -0.5
    wq, sc = ops.shadow_fp8(w)
    bias = torch.randn(N, device=d, generator=g)
    res = bf(torch.randn(M, N, device=d, generator=g))
    aux_in = bf(torch.randn(M, N, device=d, generator=g))
    v0 = a.double() @ _deq(w).t()
    vb = v0 + bias.double()
    ck = Check()
    ONE, TWO = 1.01 * U, 2.02 * U
    mm = lambda **kw: ops.mm_nt(a, wq, b_scale=sc, **kw)
    ck.tight(mm(), v0, "plain", rtol=ONE)
    assert _lib.lib().focus_gemm_last_kernel() == 2, "fp8 weights must take the wave-specialised kernel"
    ck.tight(mm(bias=bias), vb, "bias", rtol=ONE)
    ck.tight(mm(bias=bias, residual=res), vb + res.double(), "bias+residual", rtol=TWO, mag=vb)
    aux = torch.empty(M, N, device=d, dtype=torch.bfloat16)
    ck.tight(mm(bias=bias, aux=aux, epilogue=ops.EPI_GELU), gelu64(vb), "gelu", rtol=3 * U)
    ck.tight(aux, vb, "gelu saved pre-activation", rtol=ONE)
    ck.tight(mm(bias=bias, epilogue=ops.EPI_RELU, residual=res), torch.relu(vb) + res.double(), "relu+residual", rtol=TWO,
             mag=torch.relu(vb))
    x = aux_in.double()
    ck.tight(mm(aux=aux_in, epilogue=_lib.EPI_DGELU), v0 * dgelu64(x), "dgelu", rtol=TWO)
    if M * N < 20_000_000:
        ck.tight(mm(bias=bias, epilogue=ops.EPI_TANH), torch.tanh(vb), "tanh", rtol=TWO)
        ck.tight(mm(aux=aux_in, epilogue=_lib.EPI_DRELU), v0 * (x > 0), "drelu", rtol=ONE)
        ck.tight(mm(aux=aux_in, epilogue=_lib.EPI_DTANH), v0 * (1 - x * x), "dtanh", rtol=TWO)
    ck.done()


def test_fp8_b_operand_needs_the_mfma_path():
    """No silent detour: an fp8 B with a shape the kernel cannot take is an error."""
    from focus_amd import ops
    d = dev()
    a = torch.randn(256, 100, device=d).bfloat16()            # K = 100 is not a multiple of 64
    wq = torch.zeros(128, 100, dtype=torch.uint8, device=d)
    with pytest.raises(RuntimeError):
        ops.mm_nt(a, wq, b_scale=torch.ones(1, device=d))


@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_fp8_linear_and_mlp_autograd(act):
    """ops.linear / ops.mlp inside ops.fp8_weights: forward and d(input) multiply the e4m3 copy; the weight and bias
    gradients are dY^T.X as always (they are the gradients with respect to the rounded weights)."""
    from focus_amd import ops
    ops.drop_caches()
    d = dev()
    g = torch.Generator().manual_seed(3)
    M, Din, H = 2048, 768, 3072
    x = torch.randn(M, Din, generator=g).bfloat16()
    w1, b1 = torch.randn(H, Din, generator=g) * Din ** -0.5, 0.1 * torch.randn(H, generator=g)
    w2, b2 = torch.randn(Din, H, generator=g) * H ** -0.5, 0.1 * torch.randn(Din, generator=g)
    ct = torch.randn(M, Din, generator=g).bfloat16()
    P = [t.to(d).requires_grad_() for t in (w1, b1, w2, b2)]
    xg = x.to(d).requires_grad_()
    epi = ops.EPI_GELU if act == "gelu" else ops.EPI_RELU
    with ops.fp8_weights(True):
        y = ops.mlp(xg, P[0], P[1], P[2], P[3], act=epi)
        assert len(ops._fp8_cache) == 2
        y2 = ops.linear(xg, P[0], P[1])
    ((y.float() * ct.to(d).float()).sum() + y2.float().square().mean()).backward()
    # fp64 reference on the rounded weights, bf16 roundings of the intermediates replayed
    R = [_deq(P[0]).requires_grad_(), b1.double().to(d).requires_grad_(), _deq(P[2]).requires_grad_(),
         b2.double().to(d).requires_grad_()]
    xr = x.double().to(d).requires_grad_()
    z = xr @ R[0].t() + R[1]
    hcur = gelu64(z) if act == "gelu" else torch.relu(z)
    yr = hcur @ R[2].t() + R[3]
    y2r = xr @ R[0].t() + R[1]
    ((yr * ct.double().to(d)).sum() + y2r.square().mean()).backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    assert rel(y, yr.detach()) < 2 ** -6 and rel(y2, y2r.detach()) < 2 ** -7
    assert rel(xg.grad, xr.grad) < 2 ** -5
    for p, r, n in zip(P, R, ("w1", "b1", "w2", "b2")):
        assert rel(p.grad, r.grad) < 2 ** -5, n


def test_fp8_copies_follow_the_optimizer():
    """After an optimizer step the e4m3 copies (both orientations) and their scales are those of the UPDATED masters."""
    from focus_amd import ops
    from focus_amd.slowfast.models.optimizer import FusedAdamW
    from oracle import fp8
    ops.drop_caches()
    d = dev()
    w = torch.nn.Parameter((torch.randn(768, 384, generator=torch.Generator().manual_seed(1)) * 0.05).to(d))
    opt = FusedAdamW([w], lr=1e-2, weight_decay=0.0)
    x = torch.randn(2048, 384, device=d).bfloat16()
    for _ in range(2):
        with ops.fp8_weights(True):
            y = ops.linear(x, w)
        y.float().square().mean().backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    codes, sc = ops.shadow_fp8(w)
    ref_codes, ref_scale = fp8.quantize_per_tensor(w.detach().cpu().numpy())
    nz = (ref_codes & 0x7F) != 0
    assert np.array_equal(codes.cpu().numpy()[nz], ref_codes[nz]) and float(sc) == float(ref_scale)
    assert torch.equal(ops.shadow_fp8(w, transposed=True)[0], codes.t().contiguous())


def test_motionformer_hr_fp8_full_size_vs_oracle(oracle):
    """BASELINE configs[4] as stated: ORViT-Motionformer-HR 16x336, 6 objects, EK heads, fp8 (e4m3) Linear weights with
    bf16 activations, one synthetic clip, against the CPU oracle run on the SAME e4m3-rounded weights: verb / noun logits,
    both EK losses, the 8 gradients of test_gpu_hr.test_motionformer_hr_full_size_vs_oracle."""
    from focus_amd import ops
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    from focus_amd.train import synthetic_batch
    from oracle import fp8
    import bench
    ops.drop_caches()
    cfg = bench.make_cfg(1, 1, mixed=True, hr=True)
    cfg.merge_from_list(["TRAIN.FP8_WEIGHTS", True])
    torch.manual_seed(0)
    m = build_model(cfg)
    assert m.fp8_weights
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
    names = ["head0.weight", "head1.bias", "pre_logits.fc.weight", "blocks.11.mlp.fc2.weight", "blocks.11.attn.qkv.weight",
             "blocks.10.patch_to_d.2.weight", "blocks.10.attn.proj_kv.weight", "blocks.10.motion_mlp.fc1.weight"]
    params = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    quantised = sorted(k for k in params if QUANTISED.search(k))
    assert len(quantised) == 12 * 5 + 3 * 4
    for k in quantised:        # what the product multiplies by: decode(e4m3(w * 448 / amax)) * amax / 448
        params[k] = torch.from_numpy(fp8.fake_quant(params[k].numpy()))
    for k in names:
        params[k].requires_grad_()
    ocfg = dict(depth=12, heads=12, orvit_layers=[1, 6, 10], temporal_resolution=8, patch=(2, 16, 16), crop=336)
    _, ref = oracle.motionformer_forward(params, inputs[0], meta["orvit_bboxes"], ocfg, training=True)
    rl = oracle.ek_loss(ref, labels)
    ref_loss = rl["verb_loss"] + rl["noun_loss"]
    ref_loss.backward()
    d = dev()
    _, got = m([inputs[0].to(d)], {"orvit_bboxes": meta["orvit_bboxes"].to(d)})
    by_id = {id(p): n for n, p in m.named_parameters()}
    assert sorted(by_id[i] for i in ops._fp8_cache) == quantised, "the set of fp8 weights is part of the contract"
    close(got["verb"], ref["verb"], 3e-2, "HR fp8 verb logits")
    close(got["noun"], ref["noun"], 3e-2, "HR fp8 noun logits")
    ld = get_loss_func(cfg)(reduction="mean")(got, {k: v.to(d) for k, v in labels.items()})
    for k in ("verb_loss", "noun_loss"):
        assert abs(float(ld[k].detach()) - float(rl[k])) < 3e-2 * max(1.0, float(rl[k])), k
    loss = ld["verb_loss"] + ld["noun_loss"]
    loss.backward()
    named = dict(m.named_parameters())
    for k in names:
        gr = params[k].grad
        close(named[k].grad, gr, 9e-2, "HR fp8 grad " + k, floor=1e-2 * float(gr.abs().max()) + 1e-8)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
