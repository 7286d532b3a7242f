"""Kernel-level parity of the TIMED bf16 kernels (the ones bench.py measures), one kernel per check.

The model-level tests compare a bf16 pipeline with an fp32/fp64 pipeline, so input quantisation dominates and a
kernel bug that perturbs a few percent of the elements by a few percent could hide inside their 3e-2 tolerance.
Here every kernel is fed bf16-ROUNDED inputs and compared with an fp64 evaluation of the SAME values, element by
element:   |got - want| <= rtol * max(|want|, floor * max|want|)   with rtol = 2^-7 (two bf16 ulps) for single-kernel
outputs; the only slack is the output's own bf16 rounding (2^-9) and the bf16 rounding of MFMA operands the kernel forms
internally (probabilities, dL).  A wrong tail mask, swizzle or epilogue lands far outside that."""
import numpy as np
import pytest
import torch

from test_gpu_parity import dev

pytestmark = pytest.mark.gpu

U = 2.0 ** -8            # bf16 unit roundoff (8 significant bits, round to nearest even)
RT = 2.0 ** -7           # two roundings


class Check:
    """Collects every comparison of a test and reports them together (one GPU run shows all margins).

    tight(got, want, what, rtol, floor, mag):  |got - want| <= rtol * max(|want|, mag, floor * max|want|) per element.
    `mag` (optional, same shape) is the magnitude of the terms the kernel sums where it rounds an MFMA OPERAND to bf16
    internally (attention probabilities, dL, slot weights): sum_i |a_i b_i|, the scale of the classical dot-product
    error bound -- an output that cancels to ~0 still carries the operands' rounding error."""

    def __init__(self):
        self.rows, self.bad = [], []

    def tight(self, got, want, what, rtol=RT, floor=1e-2, mag=None):
        want = want.detach().double()
        got = got.detach().to(want.device).double()        # compared where the fp64 reference lives (GPU for the GEMMs)
        assert got.shape == want.shape, (what, got.shape, want.shape)
        scale = float(want.abs().max())
        den = torch.clamp(want.abs(), min=floor * scale)
        if mag is not None:
            den = torch.maximum(den, mag.detach().to(want.device).double().abs())
        ratio = float(((got - want).abs() / den).max()) if bool(torch.isfinite(got).all()) else float("inf")
        self.rows.append("%-52s %.3e (limit %.3e)" % (what, ratio, rtol))
        if not ratio <= rtol:
            self.bad.append(self.rows[-1])

    def done(self):
        print("\n".join(self.rows))
        import os
        out = os.environ.get("FOCUS_MARGINS")              # optional: keep the measured margins of a GPU run
        if out:
            with open(out, "a") as f:
                f.write(os.environ.get("PYTEST_CURRENT_TEST", "") + "\n  " + "\n  ".join(self.rows) + "\n")
        assert not self.bad, "elementwise error above the limit:\n" + "\n".join(self.bad)


def bf(t):
    return t.bfloat16()


def gelu64(v):
    return 0.5 * v * (1.0 + torch.erf(v / 2.0 ** 0.5))


def dgelu64(v):
    return 0.5 * (1.0 + torch.erf(v / 2.0 ** 0.5)) + v * torch.exp(-0.5 * v * v) / (2.0 * np.pi) ** 0.5


# ----------------------------------------------------------------------------------------------------------------
# NT GEMM: every epilogue, on the bench's own shapes and on ragged edges
# ----------------------------------------------------------------------------------------------------------------
BENCH_SHAPES = [(12552, 768, 768), (12552, 2304, 768), (12552, 3072, 768), (12552, 768, 3072), (100352, 768, 768)]
RAGGED_SHAPES = [(1601, 768, 768), (333, 200, 192), (12808, 1536, 768), (129, 97, 64), (6272, 384, 768)]
# few rows: gemm_mfma_small.hip (the recurrent STEVE Linears at B*K = 352 rows, the motion stream at 256, ragged edges)
SMALL_SHAPES = [(352, 192, 192), (352, 576, 192), (352, 192, 768), (352, 768, 192), (256, 768, 1536), (45, 72, 64),
                (1000, 40, 128), (17, 8, 1536),
                # more than one pass over the register-resident operands (K > 1536): the motion stream's fc2 and its dX
                (256, 768, 3072), (256, 768, 2304), (77, 40, 4608), (256, 3072, 768)]


@pytest.mark.parametrize("shape", BENCH_SHAPES + RAGGED_SHAPES + SMALL_SHAPES)
def test_nt_gemm_epilogues(shape):
    from focus_amd import ops
    M, N, K = shape
    d = dev()
    g = torch.Generator(device=d).manual_seed(M + N + K)
    a = bf(torch.randn(M, K, device=d, generator=g))
    w = bf(torch.randn(N, K, device=d, generator=g) * K ** -0.5)
    bias = torch.randn(N, device=d, generator=g)
    res = bf(torch.randn(M, N, device=d, generator=g))
    aux_in = bf(torch.randn(M, N, device=d, generator=g))           # saved pre-activation / output of the backward forms
    v0 = a.double() @ w.double().t()                                 # fp64 product of the same bf16 values
    big = M * N > 20_000_000
    ck = Check()
    ONE, TWO = 1.01 * U, 2.02 * U
    # The epilogue rounds z = alpha*acc + bias to bf16 ONCE (that z is what the GELU form saves for backward), applies
    # the activation / act' factor / residual to the rounded z and rounds the result: plain, bias and the ReLU mask
    # are exact to one rounding; forms that touch z again carry two (the reference's fp16 autocast has the same
    # structure: Linear output rounded, then the activation / residual add).
    vb = v0 + bias.double()
    ck.tight(ops.mm_nt(a, w), v0, "plain", rtol=ONE)
    ck.tight(ops.mm_nt(a, w, bias=bias), vb, "bias", rtol=ONE)
    ck.tight(ops.mm_nt(a, w, bias=bias, residual=res), vb + res.double(), "bias+residual", rtol=TWO, mag=vb)
    aux = torch.empty(M, N, device=d, dtype=torch.bfloat16)
    # gelu(round(z)) rounded again: (|z gelu'(z)| + |gelu(z)|) u <= 2.3 u |gelu(z)| for z > 0
    ck.tight(ops.mm_nt(a, w, bias=bias, aux=aux, epilogue=ops.EPI_GELU), gelu64(vb), "gelu", rtol=3 * U)
    ck.tight(aux, vb, "gelu saved pre-activation", rtol=ONE)
    if M >= 12000:     # wave-specialised kernel: C = gelu(saved z) exactly -- forward and backward meet at the same z
        ck.tight(ops.mm_nt(a, w, bias=bias, aux=aux, epilogue=ops.EPI_GELU), gelu64(aux.double()),
                 "gelu of the saved pre-activation", rtol=ONE)
    ck.tight(ops.mm_nt(a, w, bias=bias, epilogue=ops.EPI_RELU, residual=res), torch.relu(vb) + res.double(),
             "relu+residual", rtol=TWO, mag=torch.relu(vb))
    if not big:
        ck.tight(ops.mm_nt(a, w, bias=bias, epilogue=ops.EPI_TANH), torch.tanh(vb), "tanh", rtol=TWO)
    # backward forms: C = round(z) * act'(aux)
    from focus_amd import _lib
    x = aux_in.double()
    ck.tight(ops.mm_nt(a, w, aux=aux_in, epilogue=_lib.EPI_DGELU), v0 * dgelu64(x), "dgelu", rtol=TWO)
    if not big:
        ck.tight(ops.mm_nt(a, w, aux=aux_in, epilogue=_lib.EPI_DRELU), v0 * (x > 0), "drelu", rtol=ONE)
        ck.tight(ops.mm_nt(a, w, aux=aux_in, epilogue=_lib.EPI_DTANH), v0 * (1 - x * x), "dtanh", rtol=TWO)
        # fp32 output of the same kernel family (split-K path): no output rounding, fp32 accumulation order only
        ck.tight(ops.mm_nt(a, w, out_dtype=torch.float32), v0, "fp32 out", rtol=1e-4)
    ck.done()


@pytest.mark.parametrize("shape", [(12552, 768, 768), (12552, 768, 3072), (100352, 768, 768), (12808, 768, 1536)])
def test_dx_gemm_through_transposed_shadow(shape):
    """dX = dY . W (the backward of every Linear) through the transposed bf16 weight shadow."""
    from focus_amd import ops
    M, N, K = shape                                                  # dy [M,N], w [N,K] -> dx [M,K]
    d = dev()
    g = torch.Generator(device=d).manual_seed(7 + M + N)
    dy = bf(torch.randn(M, N, device=d, generator=g))
    w = (torch.randn(N, K, device=d, generator=g) * N ** -0.5)
    got = ops._dx_from(dy, w, torch.bfloat16)
    ck = Check()
    ck.tight(got, dy.double() @ bf(w).double(), "dx", rtol=1.01 * U)
    ck.done()


# ----------------------------------------------------------------------------------------------------------------
# trajectory attention: space step (fwd, dq, dkv, delta, cls) and time step (fwd, bwd)
# ----------------------------------------------------------------------------------------------------------------
def _space_ref(qkv, F_, P, heads, cts):
    """attention.py:509-535 in fp64 on the given values; returns outputs and d(qkv)."""
    B, N, C3 = qkv.shape
    C, S = C3 // 3, F_ * P
    d = C // heads
    q64 = qkv.double().requires_grad_()

    def sh(t):
        return t.reshape(B, N, heads, d).permute(0, 2, 1, 3)
    q, k, v = (sh(t) for t in q64.split(C, dim=-1))
    scale = d ** -0.5
    cls = (torch.softmax((q[:, :, :1] * scale) @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, 1, C)
    A = torch.softmax((q[:, :, 1:] @ k[:, :, 1:].transpose(-1, -2)).reshape(B, heads, S, F_, P) * scale, dim=-1)
    xt = torch.einsum("bhsfp,bhfpd->bhsfd", A, v[:, :, 1:].reshape(B, heads, F_, P, d))
    xt = xt.permute(0, 2, 3, 1, 4).reshape(B, S, F_, C)
    xd = xt[:, torch.arange(S), torch.arange(S) // P]
    ((xt * cts[0].double().cpu()).sum() + (xd * cts[1].double().cpu()).sum() + (cls * cts[2].double().cpu()).sum()).backward()
    with torch.no_grad():
        def mh(t):                                        # [B,h,*,d] -> [B,*,(h d)]
            return t.permute(0, 2, 1, 3).reshape(B, t.shape[2], C)
        vf = v[:, :, 1:].reshape(B, heads, F_, P, d)
        # sum_p A |v|: the scale of the error the bf16 probabilities carry into P.V
        mag = torch.einsum("bhsfp,bhfpd->bhsfd", A, vf.abs()).permute(0, 2, 3, 1, 4).reshape(B, S, F_, C)
        mag_d = mag[:, torch.arange(S), torch.arange(S) // P]
        # backward (flash form): dX = d(x~) + [own frame] d(x_diag);  dP = dX.v;  dL = A (dP - sum_p A dP) scale
        dX = cts[0].double().cpu().clone()
        dX[:, torch.arange(S), torch.arange(S) // P] += cts[1].double().cpu()
        dXh = dX.reshape(B, S, F_, heads, d).permute(0, 3, 1, 2, 4)                      # [B,h,S,F,d]
        dP = torch.einsum("bhsfd,bhfpd->bhsfp", dXh, vf)
        dL = A * (dP - (A * dP).sum(-1, keepdim=True)) * scale
        kf, qh = k[:, :, 1:].reshape(B, heads, F_, P, d), q[:, :, 1:]
        mags = {"dq": mh(torch.einsum("bhsfp,bhfpd->bhsd", dL.abs(), kf.abs())),
                "dk": mh(torch.einsum("bhsfp,bhsd->bhfpd", dL.abs(), qh.abs()).reshape(B, heads, S, d)),
                "dv": mh(torch.einsum("bhsfp,bhsfd->bhfpd", A, dXh.abs()).reshape(B, heads, S, d))}
    return xt.detach(), xd.detach(), cls.detach(), q64.grad, mag, mag_d, mags


@pytest.mark.parametrize("P", [196, 200])
def test_space_attention_kernels_tight(P):
    """traj_space_fwd / traj_delta / traj_dq / traj_dkv / cls kernels at the bench's frame sizes (F=8, d=64)."""
    from focus_amd import ops
    F_, heads, B = 8, 2, 1
    C, S = heads * 64, F_ * P
    d = dev()
    g = torch.Generator().manual_seed(P)
    qkv = bf(torch.randn(B, 1 + S, 3 * C, generator=g))
    cts = [bf(torch.randn(B, S, F_, C, generator=g)), bf(torch.randn(B, S, C, generator=g)),
           bf(torch.randn(B, 1, C, generator=g))]
    xt_r, xd_r, cls_r, dq_r, mag, mag_d, mags = _space_ref(qkv, F_, P, heads, cts)
    ck = Check()
    qg = qkv.to(d).requires_grad_()
    xt, xd, cls = ops.traj_space(qg, F_, P, heads)
    ((xt.float() * cts[0].to(d).float()).sum() + (xd.float() * cts[1].to(d).float()).sum()
     + (cls.float() * cts[2].to(d).float()).sum()).backward()
    # forward: the (un-normalised, <= 1) probabilities are rounded to bf16 as the MFMA operand of P.V: <= 2^-9 of the
    # terms' magnitude, + the output's own rounding -> 2^-8 of sum_p A|v|.  (One key too many or too few in a frame's
    # softmax -- a wrong tail mask -- moves an output by ~|v|/P = 5e-3 |v|, above this limit.)
    ck.tight(xt, xt_r, "x~ (traj_space_fwd)", rtol=U, mag=mag)
    ck.tight(xd, xd_r, "x_diag (traj_space_fwd)", rtol=U, mag=mag_d)
    ck.tight(cls, cls_r, "cls row (cls_fwd)", rtol=1.01 * U)
    # backward: dL (and P for dV) are rounded to bf16 as MFMA operands, delta uses the bf16-stored x~: the error scale
    # is the magnitude of the summed terms (sum |dL||k| ...), against which the limit is ONE unit roundoff
    gq = qg.grad.double().cpu()
    ck.tight(gq[:, 1:, :C], dq_r[:, 1:, :C], "dQ (traj_delta + traj_dq)", rtol=U, mag=mags["dq"])
    ck.tight(gq[:, 1:, C:2 * C], dq_r[:, 1:, C:2 * C], "dK (traj_dkv + cls_bwd)", rtol=U, mag=mags["dk"])
    ck.tight(gq[:, 1:, 2 * C:], dq_r[:, 1:, 2 * C:], "dV (traj_dkv + cls_bwd)", rtol=U, mag=mags["dv"])
    ck.tight(gq[:, :1], dq_r[:, :1], "cls token row of dqkv (cls_bwd)", rtol=1.01 * U)
    ck.done()


@pytest.mark.parametrize("F_", [8, 4])
def test_time_attention_kernels_tight(F_):
    """time_fwd_vec / time_bwd_vec (attention.py:538-549) at S = F*196, 12 heads of 64."""
    from focus_amd import ops
    heads, B, P = 12, 1, 196
    C, S = heads * 64, F_ * P
    d = dev()
    g = torch.Generator().manual_seed(F_)
    q2 = bf(torch.randn(B, S, C, generator=g))
    k2 = bf(torch.randn(B, S, F_, C, generator=g))
    xt = bf(torch.randn(B, S, F_, C, generator=g))
    ct = bf(torch.randn(B, S, C, generator=g))
    Q, K2, X = (t.double().requires_grad_() for t in (q2, k2, xt))
    qh = Q.reshape(B, S, heads, 64) * 64 ** -0.5
    kh = K2.reshape(B, S, F_, heads, 64)
    A = torch.softmax(torch.einsum("bshd,bsfhd->bshf", qh, kh), dim=-1)
    out_r = torch.einsum("bshf,bsfhd->bshd", A, X.reshape(B, S, F_, heads, 64)).reshape(B, S, C)
    (out_r * ct.double()).sum().backward()
    qg, kg, xg = (t.to(d).requires_grad_() for t in (q2, k2, xt))
    out = ops.traj_time(qg, kg, xg, heads)
    (out.float() * ct.to(d).float()).sum().backward()
    ck = Check()
    ck.tight(out, out_r, "time out (time_fwd_vec)", rtol=1.01 * U)       # fp32 arithmetic, one output rounding
    ck.tight(qg.grad, Q.grad, "time dq2 (time_bwd_vec)", rtol=1.01 * U)
    ck.tight(kg.grad, K2.grad, "time dk2 (time_bwd_vec)", rtol=1.01 * U)
    ck.tight(xg.grad, X.grad, "time dx~ (time_bwd_vec)", rtol=1.01 * U)
    ck.done()


# ----------------------------------------------------------------------------------------------------------------
# slot attention (steve.py:76-83) at the BASELINE shape of one (frame, iteration) call
# ----------------------------------------------------------------------------------------------------------------
def test_slot_attention_kernels_tight():
    from focus_amd import ops
    B, N, K, D = 2, 4096, 11, 192
    d = dev()
    g = torch.Generator().manual_seed(2)
    k = bf(torch.randn(B, N, D, generator=g) * D ** -0.5)
    v = bf(torch.randn(B, N, D, generator=g))
    q = bf(torch.randn(B, K, D, generator=g))
    cu, ca = bf(torch.randn(B, K, D, generator=g)), bf(torch.randn(B, N, K, generator=g) * 1e-2)
    kr, vr, qr = (t.double().requires_grad_() for t in (k, v, q))
    logits = kr @ qr.transpose(-1, -2)
    logits.retain_grad()
    av = torch.softmax(logits, dim=-1)
    av.retain_grad()
    aa = av + 1e-8
    upd = (aa / aa.sum(dim=-2, keepdim=True)).transpose(-1, -2) @ vr
    ((upd * cu.double()).sum() + (av * ca.double()).sum()).backward()
    kg, vg, qg = (t.to(d).requires_grad_() for t in (k, v, q))
    u2, a2 = ops.slot_attn_step(kg, vg, qg, 1e-8)
    ((u2.float() * cu.to(d).float()).sum() + (a2.float() * ca.to(d).float()).sum()).backward()
    ck = Check()
    ck.tight(a2, av, "slot attn_vis (slot_fwd)", rtol=1.01 * U)
    with torch.no_grad():
        w_ = aa / aa.sum(dim=-2, keepdim=True)                                  # [B,N,K] weights of the mean
        mag_u = w_.transpose(-1, -2) @ vr.abs()
        mag_v = w_ @ cu.double().abs()                                          # dv[n] = sum_k w[n,k] dupd[k]
        # d logits = a (g - sum_k a g) with the bf16-STORED a and g = d/d(attn_vis): the softmax backward cancels, so
        # the scale of its rounding error is a (|g| + sum_k a |g|), not |d logits|
        gabs = av.grad.abs()
        dlg = av.detach() * (gabs + (av.detach() * gabs).sum(-1, keepdim=True))
        mag_k = dlg @ qr.abs()
        mag_q = dlg.transpose(-1, -2) @ kr.abs()
    ck.tight(u2, upd, "slot updates (slot_fwd)", rtol=U, mag=mag_u)
    ck.tight(vg.grad, vr.grad, "slot dv (slot_bwd)", rtol=RT, mag=mag_v)
    ck.tight(kg.grad, kr.grad, "slot dk (slot_bwd)", rtol=RT, mag=mag_k)
    ck.tight(qg.grad, qr.grad, "slot dq (slot_bwd)", rtol=RT, mag=mag_q)
    ck.done()


@pytest.mark.parametrize("N,K,D", [(4096, 11, 192), (1024, 7, 64), (512, 16, 256)])
def test_slot_kernels_hand_issued_loads_equal_compiler_issued_bit_for_bit(N, K, D, monkeypatch):
    """slot_fwd_mfma_kernel / slot_bwd_defer_kernel prefetch through inline-asm loads with hand-counted waits where every row
    of a workgroup exists (FULL); FOCUS_SLOT_HAND=0 runs the same arithmetic with compiler-issued loads and hipcc's own
    s_waitcnt.  Same inputs, three iterations sharing one SlotKVGrad: every output and gradient must agree bitwise (a wait
    counted one short shows up here as soon as a load is late; focus_amd/asm_lint.py is the static side of this check)."""
    from focus_amd import ops
    B, iters = 4, 3
    d = dev()
    g = torch.Generator().manual_seed(N + K)
    k = bf(torch.randn(B, N, D, generator=g) * D ** -0.5).to(d)
    v = bf(torch.randn(B, N, D, generator=g)).to(d)
    qs = [bf(torch.randn(B, K, D, generator=g)).to(d) for _ in range(iters)]
    cus = [torch.randn(B, K, D, generator=g).to(d) for _ in range(iters)]
    cas = [(torch.randn(B, N, K, generator=g) * 1e-2).to(d) for _ in range(iters)]

    def run(hand):
        monkeypatch.setenv("FOCUS_SLOT_HAND", "1" if hand else "0")
        kg, vg = k.clone().requires_grad_(), v.clone().requires_grad_()
        qg = [q.clone().requires_grad_() for q in qs]
        acc = ops.SlotKVGrad()
        outs, loss = [], 0.0
        for i in range(iters):
            u, a = ops.slot_attn_step(kg, vg, qg[i], 1e-8, acc)
            outs += [u.detach().clone(), a.detach().clone()]
            loss = loss + (u.float() * cus[i]).sum() + (a.float() * cas[i]).sum()
        loss.backward()
        torch.cuda.synchronize()
        return outs + [kg.grad, vg.grad] + [q.grad for q in qg]

    hand, auto = run(True), run(False)
    for i, (a, b) in enumerate(zip(hand, auto)):
        assert torch.equal(a, b), "tensor %d differs between hand-issued and compiler-issued loads" % i


@pytest.mark.parametrize("iters,N,K,D", [(3, 4096, 11, 192), (1, 300, 16, 64), (4, 1000, 7, 128), (2, 77, 3, 256)])
def test_slot_kv_grad_shared_by_the_iterations_of_a_frame(iters, N, K, D):
    """d(k_t), d(v_t) when `iters` corrector iterations read the same k_t, v_t (steve.py:68-83) through one SlotKVGrad:
    the per-iteration backward writes (w, dlogits) rows, slot_kv_grad_kernel forms both gradients once.  Against fp64 on
    the same bf16 values, with the magnitude-scaled bounds of the single-iteration test summed over the iterations."""
    from focus_amd import ops
    B = 2
    d = dev()
    g = torch.Generator().manual_seed(10 * iters + K)
    k = bf(torch.randn(B, N, D, generator=g) * D ** -0.5)
    v = bf(torch.randn(B, N, D, generator=g))
    qs = [bf(torch.randn(B, K, D, generator=g)) for _ in range(iters)]
    cus = [bf(torch.randn(B, K, D, generator=g)) for _ in range(iters)]
    cas = [bf(torch.randn(B, N, K, generator=g) * 1e-2) for _ in range(iters)]
    kr, vr = k.double().requires_grad_(), v.double().requires_grad_()
    qr = [q.double().requires_grad_() for q in qs]
    loss = 0.0
    mag_v = mag_k = 0.0
    avs = []
    for i in range(iters):
        av = torch.softmax(kr @ qr[i].transpose(-1, -2), dim=-1)
        av.retain_grad()
        avs.append(av)
        aa = av + 1e-8
        w_ = aa / aa.sum(dim=-2, keepdim=True)
        loss = loss + ((w_.transpose(-1, -2) @ vr) * cus[i].double()).sum() + (av * cas[i].double()).sum()
        mag_v = mag_v + w_.detach() @ cus[i].double().abs()
    loss.backward()
    with torch.no_grad():
        for i in range(iters):
            gabs = avs[i].grad.abs()
            dlg = avs[i].detach() * (gabs + (avs[i].detach() * gabs).sum(-1, keepdim=True))
            mag_k = mag_k + dlg @ qr[i].abs()
    from focus_amd import _lib
    assert _lib.lib().focus_slot_kv_grad_ok(K, D, 1, iters)              # the deferred kernel is what runs (FOCUS_BF16 = 1)
    kg, vg = k.to(d).requires_grad_(), v.to(d).requires_grad_()
    qg = [q.to(d).requires_grad_() for q in qs]
    acc = ops.SlotKVGrad()
    lg = 0.0
    for i in range(iters):
        u2, a2 = ops.slot_attn_step(kg, vg, qg[i], 1e-8, acc)
        lg = lg + (u2.float() * cus[i].to(d).float()).sum() + (a2.float() * cas[i].to(d).float()).sum()
    lg.backward()
    ck = Check()
    # one more bf16 rounding than the fused form: w and dlogits are stored as bf16 before the product
    ck.tight(vg.grad, vr.grad, "slot dv (kv_grad)", rtol=1.5 * RT, mag=mag_v)
    ck.tight(kg.grad, kr.grad, "slot dk (kv_grad)", rtol=1.5 * RT, mag=mag_k)
    for i in range(iters):
        assert qg[i].grad is not None and torch.isfinite(qg[i].grad.float()).all()
    # and the same sums as autograd forms them from independent calls (no shared state), to bf16 accumulation accuracy
    k2, v2 = k.to(d).requires_grad_(), v.to(d).requires_grad_()
    l2 = 0.0
    for i in range(iters):
        u2, a2 = ops.slot_attn_step(k2, v2, qs[i].to(d), 1e-8)
        l2 = l2 + (u2.float() * cus[i].to(d).float()).sum() + (a2.float() * cas[i].to(d).float()).sum()
    l2.backward()
    ck.tight(vg.grad, v2.grad.double(), "dv vs per-call path", rtol=2 * RT, mag=mag_v.to(d))
    ck.tight(kg.grad, k2.grad.double(), "dk vs per-call path", rtol=2 * RT, mag=mag_k.to(d))
    ck.done()


@pytest.mark.parametrize("F_,heads,S", [(8, 12, 1568), (4, 2, 100), (16, 3, 50), (8, 16, 43), (8, 6, 33), (4, 11, 37),
                                        (16, 16, 21), (8, 1, 70)])
def test_time2_kernels_tight(F_, heads, S):
    """The k2-free temporal step (time2_logits / softmax / out, time2_dl / bwd + the two batched GEMMs over g):
    out, d(q2), d(x~), d(Wk) and the exact zeros of the dead proj_kv parts, against fp64 on the same bf16 values."""
    from focus_amd import ops
    B, dh = 2, 64
    C = heads * dh
    d = dev()
    g = torch.Generator().manual_seed(100 * F_ + heads)
    q2 = bf(torch.randn(B, S, C, generator=g))
    xt = bf(torch.randn(B, S, F_, C, generator=g))
    wkv = torch.randn(2 * C, C, generator=g) * C ** -0.5
    bkv = torch.randn(2 * C, generator=g)
    cls = bf(torch.randn(B, 1, C, generator=g))
    ct = bf(torch.randn(B, S + 1, C, generator=g))
    # fp64 reference of attention.py:537-549 on the bf16-rounded weights (what the kernels read)
    Q, X, CL = (t.double().requires_grad_() for t in (q2, xt, cls))
    W = bf(wkv).double().requires_grad_()
    Bk = bkv.double().requires_grad_()
    k2 = (X @ W[:C].t() + Bk[:C]).reshape(B, S, F_, heads, dh)
    qh = Q.reshape(B, S, heads, dh) * dh ** -0.5
    A = torch.softmax(torch.einsum("bshd,bsfhd->bshf", qh, k2), dim=-1)
    out_r = torch.cat([CL, torch.einsum("bshf,bsfhd->bshd", A, X.reshape(B, S, F_, heads, dh)).reshape(B, S, C)], 1)
    (out_r * ct.double()).sum().backward()
    def run(fn):
        qg, xg, cg = (t.to(d).requires_grad_() for t in (q2, xt, cls))
        wg, bg = wkv.to(d).requires_grad_(), bkv.to(d).requires_grad_()
        out = fn(qg, xg, wg, bg, cg, heads)
        (out.float() * ct.to(d).float()).sum().backward()
        return out, qg.grad, xg.grad, wg.grad, bg.grad, cg.grad

    def l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm())
    assert ops.traj_time2_ok(xt.to(d), heads)
    out, dq, dx, dw, db, dc = run(ops.traj_time2_block)
    out0, dq0, dx0, dw0, db0, dc0 = run(ops.traj_time_block)       # the k2 path (k2 and dk2 rounded to bf16 in HBM)
    gq, gx, gw = Q.grad, X.grad, W.grad
    with torch.no_grad():      # scale of the rounding error of u (bf16) inside the logits: sum_c |u||x~| per (s,f,h) -> via A
        mag_out = torch.cat([CL.abs(), torch.einsum("bshf,bsfhd->bshd", A, X.abs().reshape(B, S, F_, heads, dh)).reshape(B, S, C)], 1)
    ck = Check()
    # u = Wk[h]^T q2 is rounded to bf16 before the logit product (the k2 path rounds k2 instead): the logits carry
    # ~u * |their terms| of error either way, which the softmax backward (a (da - sum a da): a cancellation) amplifies.
    # So the gradients are held (a) to an L2-relative error of 1e-2 and (b) to the k2 path's own error on the same
    # inputs -- the re-association must not be less accurate than what it replaces.
    ck.tight(out, out_r, "out (time2 logits/softmax/out)", rtol=4 * U, mag=mag_out)
    ck.tight(dc, CL.grad, "d cls_out", rtol=1.01 * U)
    rows = []
    for name, new, old, ref in (("d q2", dq, dq0, gq), ("d x~", dx, dx0, gx), ("d Wk", dw[:C], dw0[:C], gw[:C]),
                                ("out", out, out0, out_r)):
        e_new, e_old = l2(new, ref), l2(old, ref)
        rows.append("%-6s L2 rel err: k2-free %.3e   k2 path %.3e" % (name, e_new, e_old))
        ck.rows.append(rows[-1])
        if not (e_new < 1e-2 and e_new < 1.5 * e_old + 1e-3):
            ck.bad.append(rows[-1])
    ck.done()
    assert float(dw[C:].abs().max()) == 0.0 and float(db.abs().max()) == 0.0      # dead v2 half, shift-invariant bias
    assert float(Bk.grad[:C].abs().max()) < 1e-9 * float(gw.abs().max()) + 1e-12   # (the reference agrees: ~0)


# ----------------------------------------------------------------------------------------------------------------
# grouped weight gradients (csrc/gemm_tn_group.hip): every dW / db of a block's Linears from one launch
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows", [12552, 4100])
def test_grouped_weight_gradients_match_fp64(rows):
    """Five Linears of a Motionformer block + a bias-free one with a longer reduction and ragged sizes: the grouped launch
    against fp64 dY^T.X / column sums of the same bf16 values.  One fp32 accumulator per output element runs through the WHOLE
    reduction (no split): the bound is the classical one for a length-M fp32 chain, |err| <= 2e-6 * sum_m |dy||x| (M 2^-24 in
    the worst case; a dropped K-step of 64 rows would be 400x that)."""
    import ctypes
    from focus_amd import _lib
    d = dev()
    g = torch.Generator(device=d).manual_seed(rows)
    shapes = [(rows, 2304, 768, True), (rows - 8, 768, 768, True), (rows, 768, 768, True), (rows, 3072, 768, True),
              (rows, 768, 3072, True), (4 * rows + 24, 384, 200, False), (rows, 8, 264, True)]
    L = _lib.lib()
    Item = _lib.WgradItem
    arr = (Item * len(shapes))()
    keep = []
    for n, (M, N, K, hb) in enumerate(shapes):
        dy = bf(torch.randn(M, N + 8, device=d, generator=g))[:, :N]          # a strided view: ld_dy = N + 8
        x = bf(torch.randn(M, K, device=d, generator=g))
        dw = torch.full((N, K), float("nan"), device=d)
        db = torch.zeros(N, device=d) if hb else None
        keep.append((dy, x, dw, db))
        arr[n].dy, arr[n].x, arr[n].dw = dy.data_ptr(), x.data_ptr(), dw.data_ptr()
        arr[n].db = db.data_ptr() if hb else None
        arr[n].ld_dy, arr[n].ld_x, arr[n].M, arr[n].N, arr[n].K = dy.stride(0), x.stride(0), M, N, K
    assert L.focus_linear_wgrad_group_units(arr, len(shapes)) == 54 + 18 + 18 + 72 + 72 + 2 * 2 + 1 * 3
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nu = L.focus_linear_wgrad_group_units(arr, len(shapes))
    host = torch.empty(nu * 8, dtype=torch.uint8)
    _lib.check(L.focus_linear_wgrad_group_plan(arr, len(shapes), ctypes.c_void_p(host.data_ptr()), nu * 8), "plan")
    rec = host.view(torch.int16).view(nu, 4)
    assert len({(int(a), int(b), int(c)) for a, b, c, _ in rec.tolist()}) == nu        # every tile exactly once
    table = host.to(d)
    _lib.check(L.focus_linear_wgrad_group(arr, len(shapes), ctypes.c_void_p(table.data_ptr()), stream), "linear_wgrad_group")
    torch.cuda.synchronize()
    ck = Check()
    for n, (dy, x, dw, db) in enumerate(keep):
        ck.tight(dw, dy.double().t() @ x.double(), "dW %d %s" % (n, shapes[n][:3]), rtol=2e-6, floor=0.0,
                 mag=dy.double().abs().t() @ x.double().abs())
        if db is not None:
            ck.tight(db, dy.double().sum(0), "db %d" % n, rtol=2e-6, floor=0.0, mag=dy.double().abs().sum(0))
    ck.done()


def test_wgrad_group_context_equals_ungrouped():
    """ops.wgrad_group around Linear + MLP nodes: same outputs and input gradients bit for bit, parameter gradients equal
    to the per-Linear path up to the fp32 summation order (the per-Linear path splits the reduction over slabs)."""
    from focus_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(9)
    M, D, H = 8192, 768, 3072
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(d).requires_grad_()
    P = [mk(3 * D, D, sc=D ** -0.5), mk(3 * D, sc=0.1), mk(H, D, sc=D ** -0.5), mk(H, sc=0.1), mk(D, H, sc=H ** -0.5), mk(D, sc=0.1)]
    x = torch.randn(M, D, generator=g).bfloat16().to(d)
    ct = torch.randn(M, D, generator=g).bfloat16().to(d)

    def run(grouped):
        for p in P:
            p.grad = None
        xg = x.clone().requires_grad_()

        def body():
            q = ops.linear(xg, P[0], P[1])
            y = ops.mlp(q[:, :D].contiguous(), P[2], P[3], P[4], P[5], residual=q[:, D:2 * D].contiguous())
            (y.float() * ct.float()).sum().backward()
            return y
        if grouped:
            with ops.wgrad_group(P):
                y = body()
        else:
            y = body()
        torch.cuda.synchronize()
        return y.detach(), xg.grad, [p.grad.clone() for p in P]

    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for a, b, name in zip(g0, g1, ("qkv.w", "qkv.b", "fc1.w", "fc1.b", "fc2.w", "fc2.b")):
        assert a.shape == b.shape
        e = float((a - b).abs().max() / a.abs().max())
        assert e < 2e-5, (name, e)


def test_time2_gw_fused_consumers_of_g():
    """csrc/traj_time2_gw.hip: dq2 = g . Wk^T per head and dWk = q2^T . g per head from ONE pass over g, against fp64 on the
    same bf16 values (dq2: one bf16 rounding of a K = 768 fp32 chain; dWk: fp32 chains over the R rows, split 21 ways)."""
    import ctypes
    from focus_amd import _lib
    d = dev()
    L = _lib.lib()
    R, heads, hd = 1568 * 2, 12, 64
    C = heads * hd
    assert L.focus_traj_time2_gw_ok(R, heads, hd, 1)
    gen = torch.Generator(device=d).manual_seed(4)
    g = bf(torch.randn(heads, R, C, device=d, generator=gen))
    q2 = bf(torch.randn(R, C, device=d, generator=gen))
    wk = bf(torch.randn(2 * C, C, device=d, generator=gen) * C ** -0.5)
    dq2 = torch.empty(R, C, device=d, dtype=torch.bfloat16)
    dwk = torch.full((C, C), float("nan"), device=d)
    nb = L.focus_traj_time2_gw_workspace_bytes(R, heads, hd)
    ws = torch.empty(nb // 4, device=d)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(L.focus_traj_time2_gw(p(g), p(q2), p(wk), C, p(dq2), p(dwk), p(ws), nb, R, heads, hd, 1,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gw")
    torch.cuda.synchronize()
    ck = Check()
    wk_h = wk[:C].double().view(heads, hd, C)
    want_dq2 = torch.einsum("hrc,hdc->rhd", g.double(), wk_h).reshape(R, C)
    ck.tight(dq2, want_dq2, "dq2", rtol=1.01 * U)
    want_dwk = torch.einsum("rhd,hrc->hdc", q2.double().view(R, heads, hd), g.double()).reshape(C, C)
    mag = torch.einsum("rhd,hrc->hdc", q2.double().abs().view(R, heads, hd), g.double().abs()).reshape(C, C)
    ck.tight(dwk, want_dwk, "dWk", rtol=2e-6, floor=0.0, mag=mag)
    ck.done()
