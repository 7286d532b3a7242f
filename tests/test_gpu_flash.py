"""csrc/flash_attn.hip through the C-ABI (ops.flash_attention): the STEVE decoder's causal self-attention
(STEVE/transformer.py:23-49, mask :131-132 / :149-151) without the [Nq, Nk] probabilities in memory, against the plain
formula in fp64 on the same bf16 values -- outputs, the three gradients, ragged lengths, the three head sizes, column-block
inputs (one projection output), and dropout with the kernels' mask rebuilt here from its definition."""
import numpy as np
import pytest
import torch

from test_gpu_parity import dev, rel, rel_l2

pytestmark = pytest.mark.gpu


def drop_keep_reference(seed, BH, Nq, Nk, thr):
    """include/focus_amd.h (focus_flash_args): keep[bh, q, k] <=> half (k & 1) of lowbias32(seed ^ bh * 0x9E3779B1 ^
    q * 0x85EBCA77 ^ (k >> 1) * 0xC2B2AE3D) >= thr."""
    M = np.uint64(0xFFFFFFFF)
    bh = (np.arange(BH, dtype=np.uint64) * np.uint64(0x9E3779B1)) & M
    q = (np.arange(Nq, dtype=np.uint64) * np.uint64(0x85EBCA77)) & M
    kp = ((np.arange(Nk, dtype=np.uint64) >> np.uint64(1)) * np.uint64(0xC2B2AE3D)) & M
    x = np.uint64(seed) ^ bh[:, None, None] ^ q[None, :, None] ^ kp[None, None, :]
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & M
    x ^= x >> np.uint64(16)
    odd = (np.arange(Nk) & 1).astype(bool)[None, None, :]
    half = np.where(odd, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return half >= np.uint64(thr)


def reference(q, k, v, heads, scale, causal, keep, keep_scale, cu):
    """fp64 on the device: (out, dq, dk, dv) of sum(out * cu)."""
    B, Nq, C = q.shape
    Nk, d = k.shape[1], C // heads
    qd, kd, vd = (t.double().detach().requires_grad_() for t in (q, k, v))
    qh = qd.view(B, Nq, heads, d).transpose(1, 2)
    kh = kd.view(B, Nk, heads, d).transpose(1, 2)
    vh = vd.view(B, Nk, heads, d).transpose(1, 2)
    att = (qh * scale) @ kh.transpose(-1, -2)
    if causal:
        att = att.masked_fill(torch.triu(torch.ones(Nq, Nk, dtype=torch.bool, device=q.device), diagonal=1), float("-inf"))
    att = torch.softmax(att, dim=-1)
    if keep is not None:
        att = att * keep.view(B, heads, Nq, Nk).double() * keep_scale
    out = (att @ vh).transpose(1, 2).reshape(B, Nq, C)
    (out * cu.double()).sum().backward()
    return out.detach(), qd.grad, kd.grad, vd.grad


CASES = [  # B, heads, Nq, Nk, d, causal, p, fused qkv
    (2, 4, 1024, 1024, 48, True, 0.0, True),       # the decoder's shape (per frame), q | k | v of one projection
    (2, 4, 1024, 1024, 48, True, 0.1, False),      # ... with the reference's dropout
    (1, 2, 300, 300, 64, False, 0.0, False),       # ragged, no mask
    (3, 3, 77, 77, 32, True, 0.25, True),          # less than one tile
    (2, 4, 200, 333, 48, False, 0.1, False),       # cross shape, ragged on both sides
    (1, 4, 640, 640, 48, True, 0.0, False),        # several query tiles, keys not a multiple of 128
]


@pytest.mark.parametrize("B,heads,Nq,Nk,d,causal,p,fused", CASES)
def test_flash_attention_against_the_formula(B, heads, Nq, Nk, d, causal, p, fused):
    from focus_amd import ops
    dv_ = dev()
    C = heads * d
    g = torch.Generator().manual_seed(Nq + 7 * d + heads)
    if fused:
        qkv = (torch.randn(B, Nq, 3 * C, generator=g) * 0.7).bfloat16().to(dv_).requires_grad_()
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    else:
        q = (torch.randn(B, Nq, C, generator=g) * 0.7).bfloat16().to(dv_).requires_grad_()
        k = (torch.randn(B, Nk, C, generator=g) * 0.7).bfloat16().to(dv_).requires_grad_()
        v = torch.randn(B, Nk, C, generator=g).bfloat16().to(dv_).requires_grad_()
    cu = torch.randn(B, Nq, C, generator=g).to(dv_)
    scale = d ** -0.5
    assert ops.flash_ok(q, k, v, heads, causal)
    seed = torch.tensor([123456789 + Nq], device=dv_, dtype=torch.int32) if p > 0 else None
    out = ops.flash_attention(q, k, v, heads, scale, causal=causal, p=p, seed=seed)
    (out.float() * cu).sum().backward()
    keep, ks = None, 1.0
    if p > 0:
        thr = ops.drop_threshold(p)
        keep = torch.from_numpy(drop_keep_reference(int(seed.item()), B * heads, Nq, Nk, thr)).to(dv_)
        ks = 65536.0 / (65536 - thr)
        frac = 1.0 - float(keep.float().mean())
        assert abs(frac - p) < 0.01, frac                                # the mask really is a Bernoulli(p) draw
    ro, rq, rk, rv = reference(q.detach(), k.detach(), v.detach(), heads, scale, causal, keep, ks, cu)
    if fused:
        gq, gk, gv = qkv.grad[..., :C], qkv.grad[..., C:2 * C], qkv.grad[..., 2 * C:]
    else:
        gq, gk, gv = q.grad, k.grad, v.grad
    # bf16 outputs of fp32 accumulations; the probabilities are rounded to bf16 before they meet v / dO (2^-9 each)
    for name, a, b in (("out", out, ro), ("dq", gq, rq), ("dk", gk, rk), ("dv", gv, rv)):
        assert torch.isfinite(a.float()).all(), name
        e2, em = rel_l2(a, b.float()), rel(a, b.float())
        assert e2 < 8e-3 and em < 4e-2, "%s: L2 %.3e max %.3e" % (name, e2, em)


def test_flash_dropout_draws_differ_by_seed_and_match_when_equal():
    from focus_amd import ops
    dv_ = dev()
    q = torch.randn(1, 256, 192, device=dv_).bfloat16()
    s1 = torch.tensor([1], device=dv_, dtype=torch.int32)
    s2 = torch.tensor([2], device=dv_, dtype=torch.int32)
    a = ops.flash_attention(q, q, q, 4, 48 ** -0.5, causal=True, p=0.1, seed=s1)
    b = ops.flash_attention(q, q, q, 4, 48 ** -0.5, causal=True, p=0.1, seed=s1)
    c = ops.flash_attention(q, q, q, 4, 48 ** -0.5, causal=True, p=0.1, seed=s2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    torch.manual_seed(5)
    d1 = ops.flash_attention(q, q, q, 4, 48 ** -0.5, causal=True, p=0.1)
    torch.manual_seed(5)
    d2 = ops.flash_attention(q, q, q, 4, 48 ** -0.5, causal=True, p=0.1)
    assert torch.equal(d1, d2)                                           # the seed comes from torch's generator
