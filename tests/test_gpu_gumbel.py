"""The row passes of csrc/gumbel.hip through the C ABI: Gumbel-softmax over the dVAE vocabulary (steve.py:262-271 with
STEVE/utils.py:47-61) and the cross entropy of the decoder head (steve.py:303-306), against the same arithmetic written with
torch operators (fp32), with supplied Exp(1) draws -- what a parity run with the reference's own draws does -- and with draws
generated in the kernel, whose hash has its numpy twin here."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from rng_twins import exp1_draws, keep_mask

pytestmark = pytest.mark.gpu

TINY = float(torch.finfo(torch.float32).tiny)


def _ops():
    from focus_amd import ops
    return ops


def torch_reference(x, e_soft, e_hard, tau, hard):
    """The oracle's gumbel_softmax (oracle/focus_oracle.py, pinned by the reference's STEVE fixture) on log_softmax(x)
    (steve.py:262-264) and the arg-max of the hard sample (steve.py:266)."""
    from oracle import focus_oracle as fo
    logp = torch.log_softmax(x, dim=-1)
    z = fo.gumbel_softmax(logp, tau, hard, -1, e_soft)
    z_hard = fo.gumbel_softmax(logp, tau, True, -1, e_hard).detach()
    return z, z_hard.argmax(-1), logp


@pytest.mark.parametrize("R,V", [(37, 64), (16, 2056), (24, 4096), (5, 8192)])
@pytest.mark.parametrize("hard", [False, True])
def test_gumbel_rows_supplied_noise_fp32(R, V, hard):
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(R * 7 + V)
    x = (3.0 * torch.randn(R, V, device=dev, generator=g)).requires_grad_(True)
    e_soft = torch.empty(R, V, device=dev).exponential_(generator=g)
    e_hard = torch.empty(R, V, device=dev).exponential_(generator=g)
    w = torch.randn(R, V, device=dev, generator=g)
    tau = 0.7
    assert ops.rows_ok(x)
    z, target = ops.gumbel_softmax_rows(x, tau, hard, e_soft, e_hard)
    (z * w).sum().backward()
    dx, x.grad = x.grad, None
    zr, tr, _ = torch_reference(x, e_soft, e_hard, tau, hard)
    (zr * w).sum().backward()
    assert torch.equal(target, tr)
    assert (z - zr).abs().max().item() < 2e-6
    if hard:
        assert torch.equal(z.argmax(-1), zr.argmax(-1)) and float(z.max()) <= 1.0 + 2e-7
    scale = x.grad.abs().max().item()
    assert (dx - x.grad).abs().max().item() < 2e-5 * scale + 1e-7


@pytest.mark.parametrize("hard", [False, True])
def test_gumbel_rows_supplied_noise_bf16(hard):
    """bf16 logits in, bf16 sample out, bf16 gradient in and out (the convolutions around it run in bf16 under
    TRAIN.MIXED_PRECISION); the arithmetic between is fp32."""
    ops = _ops()
    dev = torch.device("cuda:0")
    R, V, tau = 48, 4096, 0.5
    g = torch.Generator(device=dev).manual_seed(5)
    x = (2.0 * torch.randn(R, V, device=dev, generator=g)).bfloat16().requires_grad_(True)
    e_soft = torch.empty(R, V, device=dev).exponential_(generator=g)
    e_hard = torch.empty(R, V, device=dev).exponential_(generator=g)
    w = torch.randn(R, V, device=dev, generator=g).bfloat16()
    z, target = ops.gumbel_softmax_rows(x, tau, hard, e_soft, e_hard)
    assert z.dtype == torch.bfloat16
    (z.float() * w.float()).sum().backward()
    dx, x.grad = x.grad.float(), None
    xf = x.detach().float().requires_grad_(True)
    zr, tr, _ = torch_reference(xf, e_soft, e_hard, tau, hard)
    (zr * w.float()).sum().backward()
    assert torch.equal(target, tr)
    assert torch.equal(z, zr.bfloat16()) or (z.float() - zr).abs().max().item() < 4e-3      # one bf16 step at most
    scale = xf.grad.abs().max().item()
    assert (dx - xf.grad).abs().max().item() < 1e-2 * scale


@pytest.mark.parametrize("gen", [False, True])
def test_gumbel_rows_fp32_logits_bf16_sample(gen):
    """The form STEVE.forward uses under TRAIN.MIXED_PRECISION: fp32 logits and d(logits), the sample and its gradient in
    bf16 (what the bf16 decoder behind it reads and returns) == the fp32 kernel's sample rounded once."""
    ops = _ops()
    dev = torch.device("cuda:0")
    R, V, tau = 32, 4096, 0.6
    g = torch.Generator(device=dev).manual_seed(21)
    x = (2.0 * torch.randn(R, V, device=dev, generator=g)).requires_grad_(True)
    w = torch.randn(R, V, device=dev, generator=g).bfloat16()
    if gen:
        kw = dict(seed=torch.tensor([5, 6, 7, 8], device=dev, dtype=torch.int32))
    else:
        kw = dict(e_soft=torch.empty(R, V, device=dev).exponential_(generator=g),
                  e_hard=torch.empty(R, V, device=dev).exponential_(generator=g))
    z, target = ops.gumbel_softmax_rows(x, tau, False, out_dtype=torch.bfloat16, **kw)
    assert z.dtype == torch.bfloat16
    (z.float() * w.float()).sum().backward()
    dx, x.grad = x.grad, None
    assert dx.dtype == torch.float32
    z32, t32 = ops.gumbel_softmax_rows(x, tau, False, **kw)
    (z32 * w.float()).sum().backward()
    assert torch.equal(target, t32) and torch.equal(z, z32.bfloat16())
    assert torch.equal(dx, x.grad)               # the same bf16 gradient values enter the same arithmetic


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gumbel_rows_generated_noise(dtype):
    """Draws made in the kernel == the numpy twin of the hash; the backward regenerates the same draws."""
    ops = _ops()
    dev = torch.device("cuda:0")
    R, V, tau = 40, 4096, 0.8
    g = torch.Generator(device=dev).manual_seed(11)
    x = (2.0 * torch.randn(R, V, device=dev, generator=g)).to(dtype).requires_grad_(True)
    w = torch.randn(R, V, device=dev, generator=g)
    seed = torch.tensor([123456789, -987654321, 42, -7], device=dev, dtype=torch.int32)
    z, target = ops.gumbel_softmax_rows(x, tau, False, seed=seed)
    (z.float() * w).sum().backward()
    dx = x.grad.float()
    su = [int(v) & 0xffffffff for v in seed.tolist()]
    e_soft = torch.from_numpy(exp1_draws(su[0], su[1], R, V)).to(dev)
    e_hard = torch.from_numpy(exp1_draws(su[2], su[3], R, V)).to(dev)
    xf = x.detach().float().requires_grad_(True)
    zr, tr, _ = torch_reference(xf, e_soft, e_hard, tau, False)
    (zr * w).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    assert (z.float() - zr).abs().max().item() < tol           # hardware log2 / exp2 against libm: ~1e-6 relative
    assert int((target != tr).sum()) <= 1
    scale = xf.grad.abs().max().item()
    assert (dx - xf.grad).abs().max().item() < (1e-4 if dtype == torch.float32 else 1e-2) * scale
    # same seed -> same sample; another seed -> another sample
    z2, t2 = ops.gumbel_softmax_rows(x.detach(), tau, False, seed=seed)
    assert torch.equal(z2, z.detach()) and torch.equal(t2, target)
    z3, t3 = ops.gumbel_softmax_rows(x.detach(), tau, False, seed=seed + 1)
    assert not torch.equal(t3, target)


def test_gumbel_generated_noise_is_exponential():
    """Uniform logits: the targets are uniform over the vocabulary, the two streams are independent, and the draws have the
    moments of Exp(1)."""
    ops = _ops()
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)                       # the kernel's seed is drawn from torch's generator
    R, V = 65536, 64
    x = torch.zeros(R, V, device=dev)
    z, target = ops.gumbel_softmax_rows(x, 1.0, True)
    counts = torch.bincount(target, minlength=V).float()
    sigma = (R / V * (1 - 1 / V)) ** 0.5
    assert (counts - R / V).abs().max().item() < 5.5 * sigma
    agree = (z.argmax(-1) == target).float().mean().item()
    assert abs(agree - 1.0 / V) < 5 * (1.0 / V / R) ** 0.5 + 1e-3
    e = exp1_draws(1, 2, 4096, 256).astype(np.float64)
    assert abs(e.mean() - 1.0) < 0.01 and abs(e.var() - 1.0) < 0.03


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,V,smoothing", [(50, 4096, 0.0), (33, 64, 0.1), (7, 8192, 0.0)])
def test_xent_rows(dtype, R, V, smoothing):
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(R + V)
    logits = (2.0 * torch.randn(R, V, device=dev, generator=g)).to(dtype).requires_grad_(True)
    target = torch.randint(0, V, (R,), device=dev, generator=g)
    assert ops.rows_ok(logits)
    loss = ops.label_smoothing_ce(logits, target, smoothing)
    (loss * 3.5).backward()
    ref_in = logits.detach().float().requires_grad_(True)
    ref = F.cross_entropy(ref_in, target, label_smoothing=smoothing)
    (ref * 3.5).backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    assert logits.grad.dtype == dtype
    scale = ref_in.grad.abs().max().item()
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert (logits.grad.float() - ref_in.grad).abs().max().item() < tol * scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [True, False])
def test_dropout_add(dtype, with_res):
    """residual + dropout(y) in one pass == the same mask applied with torch operators; the backward rebuilds the mask."""
    ops = _ops()
    dev = torch.device("cuda:0")
    shape, p = (6, 1024, 192), 0.1
    g = torch.Generator(device=dev).manual_seed(3)
    y = torch.randn(shape, device=dev, generator=g).to(dtype).requires_grad_(True)
    res = torch.randn(shape, device=dev, generator=g).to(dtype).requires_grad_(True) if with_res else None
    w = torch.randn(shape, device=dev, generator=g).to(dtype)
    seed = torch.tensor([2024, -17], device=dev, dtype=torch.int32)
    out = ops.dropout_add(y, res, p, True, seed=seed)
    assert out.dtype == dtype and out.shape == y.shape
    out.backward(w)
    thr = ops.drop_threshold(p)
    su = [int(v) & 0xffffffff for v in seed.tolist()]
    keep = torch.from_numpy(keep_mask(su[0], su[1], y.numel(), thr)).to(dev).view(shape)
    inv = 65536.0 / (65536 - thr)
    ref = keep * y.detach().float() * inv + (res.detach().float() if with_res else 0.0)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert (out.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert torch.equal(out.detach() == (res.detach() if with_res else 0), ~keep) or abs(keep.float().mean().item() - 0.9) < 5e-3
    assert abs(keep.float().mean().item() - (1 - thr / 65536)) < 3e-3
    dref = keep * w.float() * inv
    assert (y.grad.float() - dref).abs().max().item() <= tol * dref.abs().max().item()
    if with_res:
        assert torch.equal(res.grad, w)
    # evaluation / p = 0: the plain sum
    assert torch.equal(ops.dropout_add(y.detach(), res.detach() if with_res else None, p, False),
                       y.detach() + res.detach() if with_res else y.detach())
