"""ORViT token plumbing kernels (csrc/orvit_tokens.hip) and the in-place RoIAlign over the residual stream, against the
torch.cat / slice formulation of the reference (ORViT/orvit.py:145-169) -- values and every gradient.  The copies are
bit-exact; the merge adds in fp32 and rounds once (reference: two bf16 roundings), so it is compared with the fp64 result
to one bf16 ulp."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,HW,O,C", [(2, 3, 5, 2, 64), (8, 8, 196, 4, 768), (1, 2, 441, 6, 128)])
def test_assemble_is_the_two_cats(dtype, B, T, HW, O, C):
    from focus_amd import ops
    g = torch.Generator().manual_seed(B * 100 + O)
    x = torch.randn(B, 1 + T * HW, C, generator=g).to(dtype).to(dev()).requires_grad_()
    obj = torch.randn(B, T, O, C, generator=g).to(dtype).to(dev()).requires_grad_()
    ct = torch.randn(B, 1 + T * (HW + O), C, generator=g).to(dtype).to(dev())
    got = ops.orvit_assemble(x, obj, T, HW)
    got.backward(ct)
    gx, go = x.grad.clone(), obj.grad.clone()
    x.grad = obj.grad = None
    want = torch.cat([x[:, :1], torch.cat([x[:, 1:].reshape(B, T, HW, C), obj], dim=2).flatten(1, 2)], dim=1)
    want.backward(ct)
    assert torch.equal(got, want)
    assert torch.equal(gx, x.grad) and torch.equal(go, obj.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_mm,drop", [(True, 0.0), (False, 0.0), (True, 0.3)])
@pytest.mark.parametrize("B,T,HW,O,C", [(2, 3, 5, 2, 64), (8, 8, 196, 4, 768)])
def test_merge_is_slice_cat_add(dtype, with_mm, drop, B, T, HW, O, C):
    from focus_amd import ops
    g = torch.Generator().manual_seed(B * 10 + O + int(with_mm))
    d = dev()
    x = torch.randn(B, 1 + T * HW, C, generator=g).to(dtype).to(d).requires_grad_()
    y = torch.randn(B, 1 + T * (HW + O), C, generator=g).to(dtype).to(d).requires_grad_()
    mm = torch.randn(B, T * HW, C, generator=g).to(dtype).to(d).requires_grad_() if with_mm else None
    ct = torch.randn(B, 1 + T * HW, C, generator=g).to(dtype).to(d)
    torch.manual_seed(5)
    got = ops.orvit_merge(x, y, mm, T, HW, drop, True)
    got.backward(ct)
    grads = [t.grad.clone() for t in (x, y) + ((mm,) if with_mm else ())]
    for t in (x, y, mm):
        if t is not None:
            t.grad = None
    # reference formulation in fp64 on the same values, with the same draws
    torch.manual_seed(5)
    scale = torch.ones(B, device=d, dtype=torch.float64)
    if drop > 0:
        u = torch.rand(B, dtype=torch.float32, device=d)
        scale = (torch.floor((1 - drop) + u) / (1 - drop)).double()
        assert 0 < float((scale == 0).float().mean()) < 1 or B < 4      # both kept and dropped samples at B = 8
    X, Y = x.detach().double().requires_grad_(), y.detach().double().requires_grad_()
    M = mm.detach().double().requires_grad_() if with_mm else None
    new_patch = Y[:, 1:].reshape(B, T, HW + O, C)[:, :, :HW].reshape(B, T * HW, C)
    if with_mm:
        new_patch = new_patch + M
    want = X + scale[:, None, None] * torch.cat([Y[:, :1], new_patch], dim=1)
    want.backward(ct.double())
    # one rounding of the result to the storage type (bf16 unit roundoff 2^-8) + fp32 arithmetic on the summands
    u_out = 2.0 ** -8 if dtype == torch.bfloat16 else 0.0

    def close(a, b, what, mag=None):
        b = b.detach()
        err = (a.detach().double() - b).abs()
        lim = 1.01 * u_out * b.abs() + 2.0 ** -22 * (mag if mag is not None else b.abs()) + 1e-30
        assert bool((err <= lim).all()), (what, float((err / lim).max()))

    gath = torch.cat([Y[:, :1], new_patch], dim=1).detach().abs()
    if with_mm:
        gath = gath + torch.cat([torch.zeros_like(M[:, :1]), M], dim=1).detach().abs()
    close(got, want, "out", mag=X.detach().abs() + scale[:, None, None] * gath)
    assert torch.equal(grads[0], ct)                               # dx is dout itself
    close(grads[1], Y.grad, "dy")
    assert float(grads[1].reshape(B, -1, C)[:, 1:].reshape(B, T, HW + O, C)[:, :, HW:].abs().max()) == 0.0
    if with_mm:
        close(grads[2], M.grad, "dmm")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roi_align_reads_the_stream_in_place(dtype):
    """Same crops and the same gradient (in the stream's shape, cls rows zero) as the dense-map call on x[:, 1:]."""
    from focus_amd import ops
    B, T, H, W, C, O = 3, 4, 14, 14, 128, 3
    d = dev()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 1 + T * H * W, C, generator=g).to(dtype).to(d).requires_grad_()
    wh = 20 + 150 * torch.rand(B * T * O, 2, generator=g)
    c = 112 + 60 * (torch.rand(B * T * O, 2, generator=g) - 0.5)
    rois = torch.cat([c - wh / 2, c + wh / 2], -1).clamp(0, 224).to(d)
    img = torch.arange(B * T, dtype=torch.int32).repeat_interleave(O).to(d)
    ct = torch.randn(B * T * O, H * W, C, generator=g).to(dtype).to(d)
    got = ops.roi_align_stream(x, rois, img, T, H, W, H, W, H / 224)
    got.backward(ct)
    gx = x.grad.clone()
    x.grad = None
    want = ops.roi_align_tokens(x[:, 1:].reshape(B * T, H * W, C), rois, img, H, W, H, W, H / 224)
    want.backward(ct)
    assert torch.equal(got, want)
    assert torch.equal(gx, x.grad)
    assert float(gx[:, 0].abs().max()) == 0.0
