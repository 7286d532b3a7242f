"""BASELINE configs[2]: the STEVE slot-attention update at its real shape (24 frames x 4096 tokens x 192 channels,
11 slots, 3 corrector iterations).  The oracle cannot run the full batch in seconds, so the full shape is held to
size-independent properties (attention rows are a softmax over the slots, finite outputs and gradients, every
parameter receives a gradient) and a B=2, T=2 slice of the same shape is compared with the CPU oracle."""
import pytest
import torch

from test_gpu_parity import close, dev, rel_l2

pytestmark = pytest.mark.gpu

N, D, K, IT = 4096, 192, 11, 3


def _module():
    from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo
    torch.manual_seed(0)
    return SlotAttentionVideo(IT, K, D, D, 4 * D, num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0)


def test_slot_attention_full_shape_properties():
    B, T = 32, 24
    m = _module().to(dev())
    g = torch.Generator(device=dev()).manual_seed(1)
    x = torch.randn(B, T, N, D, device=dev(), dtype=torch.bfloat16, generator=g).requires_grad_()
    noise = torch.randn(B, K, D, device=dev(), generator=g)
    slots, attn = m(x, noise=noise)
    assert slots.shape == (B, T, K, D) and attn.shape == (B, T, N, K)
    assert torch.isfinite(slots.float()).all() and torch.isfinite(attn.float()).all()
    rows = attn.float().sum(-1)
    assert float((rows - 1).abs().max()) < 2e-2                      # softmax over the K slots (bf16 storage)
    assert float(attn.min()) >= 0.0
    # slots differ between frames and between slots (the update really ran)
    assert float((slots[:, 1] - slots[:, 0]).float().abs().max()) > 1e-3
    (slots.float().square().mean() + attn.float()[..., 0].mean()).backward()
    assert x.grad is not None and torch.isfinite(x.grad.float()).all() and float(x.grad.float().abs().max()) > 0
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    # determinism: the same inputs and noise give bit-identical slots (no atomics on the forward path)
    with torch.no_grad():
        s2, _ = m(x.detach(), noise=noise)
    assert torch.equal(s2, slots.detach())


def test_slot_attention_baseline_shape_slice_vs_oracle(oracle):
    B, T = 2, 2
    m = _module()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, N, D, generator=g)
    noise = torch.randn(B, K, D, generator=g)
    cs, ca = torch.randn(B, T, K, D, generator=g), torch.randn(B, T, N, K, generator=g) * 1e-2
    p = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    xr = x.clone().requires_grad_()
    sr, ar = oracle.slot_attention_video(p, xr, noise, IT, 4, 1)
    ((sr * cs).sum() + (ar * ca).sum()).backward()
    m = m.to(dev())
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 6e-2)):
        m.zero_grad()
        xg = x.to(dev(), dtype).requires_grad_()
        s, a = m(xg, noise=noise.to(dev()))
        ((s.float() * cs.to(dev())).sum() + (a.float() * ca.to(dev())).sum()).backward()
        close(s, sr, tol, "slots %s" % dtype)
        close(a, ar, tol, "attn %s" % dtype)
        close(xg.grad, xr.grad, tol * 3, "dinputs %s" % dtype, floor=1e-2 * float(xr.grad.abs().max()))
        named = dict(m.named_parameters())
        for k in ("project_q.weight", "project_k.weight", "project_v.weight", "gru.weight_ih", "gru.bias_hh",
                  "mlp.0.weight", "norm_slots.weight", "slot_mu", "predictor.blocks.0.attn.proj_o.weight",
                  # the predictor's q | k | v projections run as one product (ops.linear_qkv) feeding the one-launch attention
                  "predictor.blocks.0.attn.proj_q.weight", "predictor.blocks.0.attn.proj_k.weight",
                  "predictor.blocks.0.attn.proj_v.weight", "norm_inputs.weight"):
            gr = p[k].grad
            close(named[k].grad, gr, tol * 3, "grad %s %s" % (k, dtype), floor=1e-2 * float(gr.abs().max()) + 1e-12)


# ------------------------------------------------------------------------------------------------
# STEVE.forward (steve.py:253-330) and one iteration of slot_train_epoch (tools/steve_train_net.py:57-126)
# ------------------------------------------------------------------------------------------------
def _steve_small(mixed):
    from focus_amd.slowfast.config.defaults import get_cfg
    from focus_amd.slowfast.models import MODEL_REGISTRY
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "STEVE"
    cfg.NUM_GPUS = 1
    cfg.TRAIN.MIXED_PRECISION = mixed
    s = cfg.SLOTS
    s.NUM_ITERS, s.NUM_SLOTS, s.CNN_HID_SIZE, s.SIZE, s.DIM, s.MLP_HID_SIZE, s.IMG_SIZE, s.VOCAB_SIZE = 2, 3, 16, 16, 32, 32, 16, 32
    s.NUM_PREDICTOR_BLOCKS, s.NUM_PREDICTOR_HEADS = 1, 2
    s.DECODER.DIM, s.DECODER.NUM_BLOCKS, s.DECODER.NUM_HEADS = 32, 2, 2
    return cfg, MODEL_REGISTRY.get("STEVE")(cfg)


@pytest.mark.parametrize("mixed", [False, True])
def test_steve_forward_vs_reference_fixture(mixed):
    """The registered STEVE class against the fixture the reference's STEVE produced (oracle/make_golden.py main_steve):
    recon, cross entropy, mse, attns, and 23 parameter gradients of mse + cross_entropy.  eval() switches the decoder's
    dropouts off as in the fixture; the Gumbel draws and the slot initialisation are the captured ones."""
    from conftest import load_golden
    from test_gpu_parity import check_param_grads
    a, p = load_golden("steve_forward_small")
    cfg, m = _steve_small(mixed)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected
    assert all(k.endswith("self_attn_mask") for k in missing), missing          # bool buffers: built by the ctor
    m = m.to(dev()).eval()
    d = dev()
    noise = {"gumbel_soft": torch.from_numpy(a["gumbel_soft"]).float().to(d),
             "gumbel_hard": torch.from_numpy(a["gumbel_hard"]).float().to(d),
             "slots": torch.from_numpy(a["slots_noise"]).float().to(d)}
    video = torch.from_numpy(a["video"]).to(d)
    recon, ce, mse, attns = m(video, float(a["tau"]), bool(a["hard"]), noise=noise)
    tol = 3e-2 if mixed else 1e-3
    close(recon, a["recon"], tol, "recon")
    close(attns, a["attns"], tol, "attns")
    close(ce, a["cross_entropy"], tol, "cross_entropy")
    # mixed: the dVAE's convolutions run in bf16, as under the reference's autocast (steve_train_net.py:95); its Gumbel-softmax
    # arithmetic is fp32 in both modes
    close(mse, a["mse"], 5e-3 if mixed else 1e-3, "mse")
    (mse + ce).backward()
    # bf16: the CNN / encoder gradients come back through the decoder's cross-attention, the slot projection and
    # T x iterations slot updates (GRU + softmax over slots) in bf16 storage: the deep-chain factor of the module header
    check_param_grads(m, a, 1e-1 if mixed else 2e-3)


def test_slot_train_step_runs_the_reference_schedule():
    """Two iterations of the slot loop: learning rates follow set_slot_lr (optimizer.py:213-222) with the warm-up /
    half-life factors of steve_train_net.py:68-82, tau follows cosine_anneal, every parameter group moves."""
    import math
    from focus_amd.slowfast.models.optimizer import construct_optimizer_slot
    from focus_amd.train import slot_train_step
    cfg, m = _steve_small(False)
    cfg.SOLVER.OPTIMIZING_METHOD = "adam"
    cfg.SOLVER.CLIP_GRAD_L2NORM = 1.0
    cfg.SLOTS_OPTIM.WARMUP_STEPS, cfg.SLOTS_OPTIM.TAU_STEPS, cfg.SLOTS_OPTIM.HALF_LIFE = 10, 20, 50
    m = m.to(dev()).train()
    opt = construct_optimizer_slot(m, cfg)
    ntrain = sum(1 for p in m.parameters() if p.requires_grad)
    assert len(opt.param_groups) == 3 and sum(len(g["params"]) for g in opt.param_groups) >= ntrain
    before = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    g = torch.Generator().manual_seed(0)
    video = torch.rand(2, 2, 3, 16, 16, generator=g).to(dev())
    losses = []
    for step in range(2):
        loss, mse, ce, recon, attns, tau = slot_train_step(m, opt, video, step, cfg)
        losses.append(float(loss))
        assert math.isfinite(losses[-1]) and recon.shape == video.shape and attns.shape == (2, 2, 3, 3, 16, 16)
        decay = math.exp(step / 50 * math.log(0.5))
        warm = (step + 1) / 10
        assert opt.param_groups[0]["lr"] == cfg.SLOTS_OPTIM.DVAE
        assert abs(opt.param_groups[1]["lr"] - decay * warm * cfg.SLOTS_OPTIM.ENC) < 1e-12
        assert abs(opt.param_groups[2]["lr"] - decay * warm * cfg.SLOTS_OPTIM.DEC) < 1e-12
        assert abs(tau - (0.45 * math.cos(math.pi * step / 20) + 0.55)) < 1e-12
    moved = {n.split(".")[0] for n, p in m.named_parameters() if p.requires_grad and not torch.equal(p.detach(), before[n])}
    assert moved == {"dvae", "steve_encoder", "steve_decoder"}


# ------------------------------------------------------------------------------------------------
# the composite nodes of the slot loop, each against plain torch in fp64 on the same bf16 values
# ------------------------------------------------------------------------------------------------
def _rel(got, want):
    want = want.double()
    return float((got.double().cpu() - want.cpu()).abs().max() / want.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("paired", [False, True])
def test_linear_kv_matches_two_linears(paired):
    """ops.linear_kv (steve.py:62-63): k = alpha x.Wk^T, v = x.Wv^T and the fused d(x) = alpha dk.Wk + dv.Wv.  `paired`: the
    cotangents arrive as the two halves of one [rows, 2D] buffer (what ops.slot_attn_step's deferred k/v gradient hands
    over) and the backward runs one product per gradient; otherwise two products with the residual epilogue."""
    from focus_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(11)
    R, Din, D, alpha = 4608, 192, 192, 192 ** -0.5
    x = torch.randn(2, R // 2, Din, generator=g).bfloat16()
    wk, wv = torch.randn(D, Din, generator=g) * Din ** -0.5, torch.randn(D, Din, generator=g) * Din ** -0.5
    ck, cv = torch.randn(2, R // 2, D, generator=g).bfloat16(), torch.randn(2, R // 2, D, generator=g).bfloat16()
    xr = x.double().requires_grad_()
    wkr, wvr = wk.bfloat16().double().requires_grad_(), wv.bfloat16().double().requires_grad_()
    kr, vr = alpha * xr @ wkr.t(), xr @ wvr.t()
    ((kr * ck.double()).sum() + (vr * cv.double()).sum()).backward()
    xg = x.to(d).requires_grad_()
    wkg, wvg = wk.to(d).requires_grad_(), wv.to(d).requires_grad_()
    k, v = ops.linear_kv(xg, wkg, wvg, alpha_k=alpha)
    if paired:
        both = torch.cat([ck, cv], dim=-1).to(d)
        torch.autograd.backward([k, v], [both[..., :D], both[..., D:]])
    else:
        torch.autograd.backward([k, v], [ck.to(d), cv.to(d)])
    assert _rel(k, kr.detach()) < 2 ** -7 and _rel(v, vr.detach()) < 2 ** -7
    assert _rel(xg.grad, xr.grad) < 2 ** -6                         # a sum of two bf16-rounded products (paired: one)
    assert _rel(wkg.grad, wkr.grad) < 2 ** -7 and _rel(wvg.grad, wvr.grad) < 2 ** -7


@pytest.mark.parametrize("defer", [False, True])
def test_gru_cell_batched_matches_torch(defer):
    """ops.gru_cell on the slot shapes (STEVE/utils.py:107-118): the batch-2 products + bias-adding gate kernel against
    torch.nn.GRUCell in fp64; with `defer` the weight / bias gradients come from ops.deferred_wgrads' flush."""
    from focus_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(12)
    R, D = 352, 192
    cell = torch.nn.GRUCell(D, D).double()
    with torch.no_grad():
        for p in cell.parameters():
            p.copy_(p.float().bfloat16().double() if p.dim() == 2 else p)
    x, h = torch.randn(R, D, generator=g).bfloat16(), torch.randn(R, D, generator=g).bfloat16()
    ct = torch.randn(R, D, generator=g).bfloat16()
    xr, hr = x.double().requires_grad_(), h.double().requires_grad_()
    (cell(xr, hr) * ct.double()).sum().backward()
    P = {n: p.detach().float().to(d).requires_grad_() for n, p in cell.named_parameters()}
    xg, hg = x.to(d).requires_grad_(), h.to(d).requires_grad_()

    def run():
        out = ops.gru_cell(xg, hg, P["weight_ih"], P["weight_hh"], P["bias_ih"], P["bias_hh"])
        (out.float() * ct.to(d).float()).sum().backward()
        return out

    if defer:
        with ops.deferred_wgrads():
            out = run()
    else:
        out = run()
    torch.cuda.synchronize()
    assert _rel(out, cell(xr, hr).detach()) < 2 ** -6
    assert _rel(xg.grad, xr.grad) < 2 ** -5 and _rel(hg.grad, hr.grad) < 2 ** -5      # gate derivatives from bf16 pre-activations
    for n, p in cell.named_parameters():
        assert P[n].grad is not None, n
        assert _rel(P[n].grad, p.grad) < 2 ** -5, n


def test_layer_norm_deferred_affine_gradients():
    """Inside ops.deferred_wgrads the LayerNorm backward leaves only block partials (focus_layernorm_bwd with
    dgamma = dbeta = NULL); the flush sums all applications at once.  Three applications of one LayerNorm against torch."""
    from focus_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(13)
    R, D = 352, 192
    w, b = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    xs = [torch.randn(R, D, generator=g).bfloat16() for _ in range(3)]
    cs = [torch.randn(R, D, generator=g) for _ in range(3)]
    wr, br = w.double().requires_grad_(), b.double().requires_grad_()
    xrs = [x.double().requires_grad_() for x in xs]
    sum((torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5) * c.double()).sum() for xr, c in zip(xrs, cs)).backward()
    wg, bg = w.to(d).requires_grad_(), b.to(d).requires_grad_()
    xgs = [x.to(d).requires_grad_() for x in xs]
    with ops.deferred_wgrads():
        sum((ops.layer_norm(xg, wg, bg, 1e-5).float() * c.to(d)).sum() for xg, c in zip(xgs, cs)).backward()
    torch.cuda.synchronize()
    assert _rel(wg.grad, wr.grad) < 2 ** -6 and _rel(bg.grad, br.grad) < 2 ** -6
    for xg, xr in zip(xgs, xrs):
        assert _rel(xg.grad, xr.grad) < 2 ** -6


def _three_linear_applications(ops, d, w, b, xs, cs, use=(0, 1, 2)):
    with ops.deferred_wgrads():
        ys = [ops.linear(x, w, b) for x in xs]
        sum((ys[i].float() * cs[i].to(d)).sum() for i in use).backward()


def test_deferred_wgrads_survive_failed_backward_frozen_params_and_partial_backward():
    """(1) a backward that raises after some applications were stashed must not poison the next step;
    (2) a frozen parameter gets no .grad from the deferred path; (3) a loss that reaches only some applications of a
    weight still yields their (partial) gradient -- the end-of-backward callback adds the leftovers."""
    from focus_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(21)
    R, Din, Dout = 96, 192, 128
    w0, b0 = 0.1 * torch.randn(Dout, Din, generator=g), 0.1 * torch.randn(Dout, generator=g)
    xs = [torch.randn(R, Din, generator=g).bfloat16() for _ in range(3)]
    cs = [torch.randn(R, Dout, generator=g) for _ in range(3)]

    def reference(use):
        wr = w0.bfloat16().double().requires_grad_()
        br = b0.double().requires_grad_()
        sum(((x.double() @ wr.t() + br) * cs[i].double()).sum() for i, x in enumerate(xs) if i in use).backward()
        return wr.grad, br.grad

    w, b = w0.to(d).requires_grad_(), b0.to(d).requires_grad_()
    xg = [x.to(d) for x in xs]

    # (1) a backward that dies half way: the hook on the second application's output raises after the third was stashed
    class Boom(RuntimeError):
        pass

    def boom(_g):
        raise Boom()
    with ops.deferred_wgrads():
        ys = [ops.linear(x, w, b) for x in xg]
        ys[1].register_hook(boom)
        with pytest.raises(Boom):
            sum((y.float() * c.to(d)).sum() for y, c in zip(ys, cs)).backward()
    w.grad = b.grad = None
    _three_linear_applications(ops, d, w, b, xg, cs)
    torch.cuda.synchronize()
    gw, gb = reference((0, 1, 2))
    assert w.grad is not None and _rel(w.grad, gw) < 2 ** -6 and _rel(b.grad, gb) < 2 ** -6

    # (2) frozen weight: no gradient appears on it, the bias still gets its own
    w.grad = b.grad = None
    w.requires_grad_(False)
    _three_linear_applications(ops, d, w, b, xg, cs)
    assert w.grad is None and b.grad is not None and _rel(b.grad, gb) < 2 ** -6
    w.requires_grad_(True)

    # (3) only applications 0 and 2 reach the loss
    w.grad = b.grad = None
    _three_linear_applications(ops, d, w, b, xg, cs, use=(0, 2))
    torch.cuda.synchronize()
    gw, gb = reference((0, 2))
    assert w.grad is not None and _rel(w.grad, gw) < 2 ** -6 and _rel(b.grad, gb) < 2 ** -6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,heads,N,M,d", [(32, 4, 11, 11, 48), (3, 3, 5, 16, 64), (2, 1, 16, 1, 8)])
def test_small_attention_one_launch(dtype, B, heads, N, M, d):
    """ops.small_attention on a handful of tokens (the slot predictor's 11 x 11, transformer.py:4-49) takes the one-launch
    kernels of small_attn.hip; against softmax attention in fp64 on the same (rounded) inputs."""
    from focus_amd import _lib, ops
    assert _lib.lib().focus_small_attn_ok(N, M, d)
    dv_ = dev()
    g = torch.Generator().manual_seed(B + N + M)
    C = heads * d
    q, k, v = (torch.randn(B, n, C, generator=g).to(dtype) for n in (N, M, M))
    ct = torch.randn(B, N, C, generator=g).to(dtype)
    scale = d ** -0.5
    qr, kr, vr = (t.double().requires_grad_() for t in (q, k, v))

    def heads_of(t):
        return t.view(t.shape[0], t.shape[1], heads, d).transpose(1, 2)

    att = torch.softmax(scale * heads_of(qr) @ heads_of(kr).transpose(-1, -2), dim=-1)
    ref = (att @ heads_of(vr)).transpose(1, 2).reshape(B, N, C)
    (ref * ct.double()).sum().backward()
    qg, kg, vg = (t.to(dv_).requires_grad_() for t in (q, k, v))
    out = ops.small_attention(qg, kg, vg, heads, scale)
    (out.float() * ct.to(dv_).float()).sum().backward()
    tol = 2 ** -6 if dtype == torch.bfloat16 else 1e-5
    assert _rel(out, ref.detach()) < tol
    for got, want in ((qg.grad, qr.grad), (kg.grad, kr.grad), (vg.grad, vr.grad)):
        assert _rel(got, want) < tol


@pytest.mark.parametrize("defer", [False, True])
def test_linear_qkv_feeds_the_one_launch_attention_in_place(defer):
    """ops.linear_qkv -> ops.small_attention on the predictor's shapes (transformer.py:33-49): q | k | v are the column
    blocks of one [rows, 3C] product, the attention reads them in place and returns its three gradients as the blocks of one
    matrix again, so d(x) is one product.  Against fp64 on the same bf16 values, weight gradients immediate and deferred."""
    from focus_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(21)
    B, N, C, heads = 32, 11, 192, 4
    x = torch.randn(B, N, C, generator=g).bfloat16()
    ws = [torch.randn(C, C, generator=g) * C ** -0.5 for _ in range(3)]
    ct = torch.randn(B, N, C, generator=g).bfloat16()
    scale = (C // heads) ** -0.5
    xr = x.double().requires_grad_()
    wr = [w.bfloat16().double().requires_grad_() for w in ws]

    def split(t):
        return t.view(B, N, heads, C // heads).transpose(1, 2)

    qr, kr, vr = (xr @ w.t() for w in wr)
    att = torch.softmax(scale * split(qr) @ split(kr).transpose(-1, -2), dim=-1)
    ref = (att @ split(vr)).transpose(1, 2).reshape(B, N, C)
    (ref * ct.double()).sum().backward()
    xg = x.to(d).requires_grad_()
    wg = [w.to(d).requires_grad_() for w in ws]

    def run():
        q, k, v = ops.linear_qkv(xg, *wg)
        assert q.stride() == (N * 3 * C, 3 * C, 1) and k.data_ptr() == q.data_ptr() + 2 * C       # column blocks of one buffer
        out = ops.small_attention(q, k, v, heads, scale)
        (out.float() * ct.to(d).float()).sum().backward()
        return out

    if defer:
        with ops.deferred_wgrads():
            out = run()
    else:
        out = run()
    torch.cuda.synchronize()
    assert _rel(out, ref.detach()) < 2 ** -6
    assert _rel(xg.grad, xr.grad) < 2 ** -5
    for got, want in zip(wg, wr):
        assert got.grad is not None and _rel(got.grad, want.grad) < 2 ** -5


def test_graphed_step_refuses_live_outputs_and_replays_bit_exact():
    """GraphedStep (focus_amd/train.py): the r2 crash came from capturing while an output of an earlier eager step was
    still referenced.  The constructor now refuses that state (raises BEFORE the capture starts); with the reference
    dropped it captures, and a replay reproduces the eager step bit for bit."""
    from focus_amd import ops
    from focus_amd.train import GraphedStep
    d = dev()
    g = torch.Generator().manual_seed(5)
    w = (0.1 * torch.randn(128, 192, generator=g)).to(d).requires_grad_()
    b = torch.zeros(128, device=d, requires_grad=True)
    x = torch.randn(256, 192, generator=g).bfloat16().to(d)

    def reset():
        w.grad = b.grad = None

    def fn():
        y = ops.linear(x, w, b)
        y.float().square().mean().backward()
        return y

    reset()
    y_eager = fn()
    ref_y, ref_g = y_eager.detach().clone(), w.grad.detach().clone()
    with pytest.raises(RuntimeError, match="still referenced"):
        GraphedStep(fn, reset=reset)                    # y_eager (grad_fn alive) is still held by this frame
    del y_eager
    gs = GraphedStep(fn, reset=reset)
    y = gs.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref_y) and torch.equal(w.grad, ref_g)


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_graphed_slot_update_training_steps_equal_eager_bit_for_bit(dropout):
    """SLOTS.GRAPH_SLOT_UPDATE: slot_train_step with the slot update replayed from captured HIP graphs
    (torch.cuda.make_graphed_callables inside STEVE._slots) against the eager step, from identical parameters, optimizer
    state and random streams: three optimizer steps, every parameter bit for bit.  Covers what the r2 graph replay did not:
    the optimizer moving the weights between replays (shadows and stacked operands refreshed in place)."""
    from focus_amd import ops
    from focus_amd.slowfast.models.optimizer import construct_optimizer_slot
    from focus_amd.train import slot_train_step
    results = []
    g = torch.Generator().manual_seed(0)
    video = torch.rand(2, 3, 3, 16, 16, generator=g).to(dev())
    for graphed in (False, False, True):
        ops.drop_caches()
        cfg, m = _steve_small(True)
        cfg.SOLVER.OPTIMIZING_METHOD = "adam"
        cfg.SOLVER.CLIP_GRAD_L2NORM = 1.0
        cfg.SLOTS.PREDICTOR_DROPOUT = dropout
        cfg.SLOTS.GRAPH_SLOT_UPDATE = graphed
        torch.manual_seed(3)
        from focus_amd.slowfast.models import MODEL_REGISTRY
        m = MODEL_REGISTRY.get("STEVE")(cfg).to(dev()).train()
        assert m.graph_slot_update == graphed
        opt = construct_optimizer_slot(m, cfg)
        torch.manual_seed(11)
        losses = []
        for step in range(3):
            loss, *_ = slot_train_step(m, opt, video, step, cfg)
            losses.append(float(loss))
        torch.cuda.synchronize()
        results.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
        if graphed:
            assert len(m._savi_graphs) == 1
    (l0, p0), (l0b, p0b), (l1, p1) = results
    conv = lambda n: "dvae." in n or ".cnn." in n or ".pos." in n       # parameters whose gradients come from MIOpen convolutions
    # Two EAGER runs first: they give the run-to-run spread of the step itself.  Sums formed with fp32 atomics (MIOpen's
    # convolution weight gradients; the split-K fallback of the narrow layers of this reduced model) depend on the order in which
    # workgroups retire, which varies from run to run -- more so since part of the step runs on a side stream -- and Adam turns
    # a flipped sign of a tiny gradient into a 2 lr step.  The graphed run must be INSIDE that spread: bit-identical wherever
    # the two eager runs are, and no further from them than they are from each other elsewhere.  (With dropout the
    # predictor's attention takes the unfused path; the random streams agree: a different mask would move the loss in the
    # second digit, not the eighth.)
    assert l0[0] == l0b[0] == l1[0], (l0, l0b, l1)                   # the first step: identical parameters and draws
    assert all(abs(a - b) <= 4 * abs(a - c) + 1e-6 * abs(a) for a, b, c in zip(l0, l1, l0b)), (l0, l0b, l1)
    exact = 0
    for n in p0:
        if not p0[n].is_floating_point():                             # (the decoder's boolean mask buffers)
            assert torch.equal(p0[n], p1[n]), n
            continue
        spread = float((p0[n] - p0b[n]).abs().max())
        if spread == 0.0 and not conv(n):                             # (MIOpen's gradients may agree twice and still differ a third time)
            assert torch.equal(p0[n], p1[n]), n
            exact += 1
        else:
            assert float((p0[n] - p1[n]).abs().max()) <= max(10 * spread, 2.5e-3), n
    assert exact >= len(p0) // 4, (exact, len(p0))                  # a good part of the model IS reproducible bit for bit


@pytest.mark.parametrize("staged", [1, 0])
def test_fused_slot_tail_matches_the_unfused_loop(monkeypatch, staged):
    """ops.slot_tail (csrc/slot_tail.hip: GRU -> LN -> MLP -> LN -> q per iteration in one launch) against the same
    SlotAttentionVideo with FOCUS_SLOT_TAIL=0 (eight launches per iteration): slots, attention maps, input gradient and every
    parameter gradient.  The two differ only in where bf16 roundings fall (the fused kernel rounds less often)."""
    from focus_amd import ops
    from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo
    d = dev()
    B, T, N, D, K = 4, 3, 512, 192, 11
    torch.manual_seed(0)
    m = SlotAttentionVideo(3, K, D, D, 4 * D, num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0).to(d)
    g = torch.Generator(device=d).manual_seed(1)
    x0 = torch.randn(B, T, N, D, device=d, generator=g).bfloat16()
    noise = torch.randn(B, K, D, device=d, generator=g)
    cs = torch.randn(B, T, K, D, device=d, generator=g)

    monkeypatch.setenv("FOCUS_SLOT_TAIL_STAGED", str(staged))         # 1: four right-sized launches, 0: one launch per tail

    def run(fused, fp32=False):
        monkeypatch.setenv("FOCUS_SLOT_TAIL", "1" if fused else "0")
        ops.drop_caches()
        for p in m.parameters():
            p.grad = None
        x = (x0.float() if fp32 else x0.clone()).requires_grad_()
        slots, attn = m(x, noise=noise)
        ((slots.float() * cs).sum() + attn.float().square().sum()).backward()
        torch.cuda.synchronize()
        return slots.detach().float(), attn.detach().float(), x.grad.float(), {n: p.grad.clone() for n, p in m.named_parameters()}

    sr, ar, dxr, gr = run(False, fp32=True)          # the same module on fp32 inputs: fp32 kernels throughout, no fused tail
    s0, a0, dx0, g0 = run(False)
    s1, a1, dx1, g1 = run(True)
    # two bf16 pipelines that round at different places, nine chained slot updates with sharp softmaxes in between: each is
    # compared with the fp32 run (L2-relative), and the fused one, which rounds less often, may not be further from it than the
    # unfused one is (25 % slack); both are held to the fp64 oracle separately (test_slot_attention_slice_vs_oracle, fused path)

    def both(f1, f0, ref, cap, what, floor=1e-3):
        e1, e0 = rel_l2(f1, ref, floor), rel_l2(f0, ref, floor)
        assert e1 < cap and e1 <= 1.25 * e0 + 5e-3, "%s: fused %.3e, unfused %.3e (cap %.1e)" % (what, e1, e0, cap)

    both(s1, s0, sr, 3e-2, "slots")
    both(a1, a0, ar, 3e-2, "attn")
    both(dx1, dx0, dxr, 1.5e-1, "d inputs")       # measured: 8.4e-2 fused, 8.6e-2 unfused
    for n in g0:
        assert g1[n] is not None, n
        if float(gr[n].abs().max()) < 1e-3 * float(g0[n].abs().max()):
            # exactly zero in exact arithmetic (norm_slots.bias shifts every slot's q alike and the softmax over the slots
            # does not see it): both bf16 runs hold rounding noise only, of the same size
            assert float(g1[n].abs().max()) <= 2.0 * float(g0[n].abs().max()) + 1e-12, n
            continue
        both(g1[n].float(), g0[n].float(), gr[n].float(), 1.5e-1, "grad " + n, floor=1e-2 * float(gr[n].abs().max()) + 1e-12)


def test_decoder_at_full_width_runs_flash_attention_and_matches_the_oracle(oracle, monkeypatch):
    """TransformerDecoder at the width of the BASELINE shape (d_model 192, 4 heads of 48, STEVE/transformer.py:117-193) in
    bf16: the causal self-attention and the cross-attention to the slots both take ops.flash_attention (no [T, S]
    probabilities in memory); output, input gradients and parameter gradients against oracle.transformer_decoder in fp32 on
    the CPU (the oracle's decoder is pinned by the reference's steve_forward_small fixture at d_model 32)."""
    from focus_amd import ops
    from focus_amd.slowfast.models.STEVE.transformer import TransformerDecoder
    d = dev()
    B, T, K, D, H, NB = 3, 320, 11, 192, 4, 2
    torch.manual_seed(3)
    m = TransformerDecoder(NB, T, D, H, dropout=0.0).to(d)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(B, T, D, generator=g)
    e0 = torch.randn(B, K, D, generator=g)
    cu = torch.randn(B, T, D, generator=g)
    calls = []
    real = ops.flash_attention
    monkeypatch.setattr(ops, "flash_attention", lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    x = x0.bfloat16().to(d).requires_grad_()
    e = e0.bfloat16().to(d).requires_grad_()
    y = m(x, e)
    (y.float() * cu.to(d)).sum().backward()
    assert len(calls) == 2 * NB, calls                                 # self- and cross-attention of every block
    p = {"tf." + k: v.detach().float().cpu() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    leaves = {k: v.clone().requires_grad_() for k, v in p.items()}
    xr = x0.bfloat16().float().requires_grad_()
    er = e0.bfloat16().float().requires_grad_()
    yr = oracle.transformer_decoder(leaves, "tf", xr, er, H, NB)
    (yr * cu).sum().backward()
    close(y, yr, 3e-2, "decoder output")
    close(x.grad, xr.grad, 5e-2, "d input")
    close(e.grad, er.grad, 5e-2, "d slots")
    named = dict(m.named_parameters())
    for k, v in leaves.items():
        n = k[3:]
        if n in named and v.grad is not None:
            # (LayerNorm gains and shifts: sums of 960 signed bf16-rounded products that largely cancel)
            close(named[n].grad, v.grad, 1e-1 if "layer_norm" in n else 6e-2, "grad " + n,
                  floor=1e-2 * float(v.grad.abs().max()) + 1e-12)


def test_shared_gradient_nodes_refuse_a_loss_on_a_subset_of_their_uses():
    """FrameGrad / SlotKVGrad hand autograd one gradient from the node that runs last.  A loss that leaves some of the counted
    nodes out of the backward pass (here: a loss on one of three frames, on one of two iterations) must raise at the end of
    that pass instead of dropping d(inputs) for the whole video; the complete loss right after still works."""
    from focus_amd import ops
    d = dev()
    B, T, N, D, K = 2, 3, 256, 64, 5
    g = torch.Generator(device=d).manual_seed(0)
    video = torch.randn(B, T, N, D, device=d, generator=g).bfloat16().requires_grad_()
    gamma = torch.ones(D, device=d, requires_grad=True)
    beta = torch.zeros(D, device=d, requires_grad=True)
    sh = ops.FrameGrad()
    ys = [ops.layer_norm_frame(video, t, gamma, beta, 1e-5, sh) for t in range(T)]
    with pytest.raises(RuntimeError, match="not reached by backward"):
        ys[1].float().sum().backward()
    video.grad = None
    sh = ops.FrameGrad()
    ys = [ops.layer_norm_frame(video, t, gamma, beta, 1e-5, sh) for t in range(T)]
    sum(y.float().square().sum() for y in ys).backward()
    assert video.grad is not None and float(video.grad.float().abs().sum()) > 0

    k = torch.randn(B, N, D, device=d, generator=g).bfloat16().requires_grad_()
    v = torch.randn(B, N, D, device=d, generator=g).bfloat16().requires_grad_()
    qs = [torch.randn(B, K, D, device=d, generator=g).bfloat16().requires_grad_() for _ in range(2)]
    acc = ops.SlotKVGrad()
    outs = [ops.slot_attn_step(k, v, q, 1e-8, acc) for q in qs]
    with pytest.raises(RuntimeError, match="not reached by backward"):
        outs[0][0].float().sum().backward()
    acc = ops.SlotKVGrad()
    outs = [ops.slot_attn_step(k, v, q, 1e-8, acc) for q in qs]
    sum(u.float().sum() + a.float().square().sum() for u, a in outs).backward()
    assert k.grad is not None and v.grad is not None


def test_fused_backward_tail_matches_the_composed_one(monkeypatch):
    """focus_slot_tail_bwd (one launch: dX chain of q-projection, LayerNorms, MLP and GRU on 16-row workgroups) against the
    backward composed of the single kernels, behind the SAME fused forward: input gradients and every parameter gradient of a
    three-frame slot loop.  Both round to bf16 at the same places; the products differ in accumulation order only."""
    from focus_amd import ops
    from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo
    d = dev()
    B, T, N, D, K = 4, 3, 512, 192, 11
    torch.manual_seed(0)
    m = SlotAttentionVideo(3, K, D, D, 4 * D, num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0).to(d)
    g = torch.Generator(device=d).manual_seed(1)
    x0 = torch.randn(B, T, N, D, device=d, generator=g).bfloat16()
    noise = torch.randn(B, K, D, device=d, generator=g)
    cs = torch.randn(B, T, K, D, device=d, generator=g)

    def run(fused_bwd):
        monkeypatch.setattr(ops, "_SLOT_TAIL_BWD", fused_bwd)
        ops.drop_caches()
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        slots, attn = m(x, noise=noise)
        ((slots.float() * cs).sum() + attn.float().square().sum()).backward()
        torch.cuda.synchronize()
        return slots.detach().float(), x.grad.float(), {n: p.grad.clone().float() for n, p in m.named_parameters()}

    s0, dx0, g0 = run(False)
    s1, dx1, g1 = run(True)
    assert torch.equal(s0, s1)                                         # the same forward
    assert rel_l2(dx1, dx0) < 2e-2, rel_l2(dx1, dx0)
    for n in g0:
        scale = float(g0[n].abs().max())
        if scale < 1e-3 * max(float(v.abs().max()) for v in g0.values()) and "norm_slots.bias" in n:
            continue                                                    # exactly zero in exact arithmetic: rounding noise
        assert rel_l2(g1[n], g0[n], floor=1e-2 * scale + 1e-12) < 3e-2, (n, rel_l2(g1[n], g0[n], floor=1e-2 * scale + 1e-12))
