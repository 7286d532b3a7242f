"""BASELINE configs[2]: the STEVE slot-attention update at its real shape (24 frames x 4096 tokens x 192 channels,
11 slots, 3 corrector iterations).  The oracle cannot run the full batch in seconds, so the full shape is held to
size-independent properties (attention rows are a softmax over the slots, finite outputs and gradients, every
parameter receives a gradient) and a B=2, T=2 slice of the same shape is compared with the CPU oracle."""
import pytest
import torch

from test_gpu_parity import close, dev

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
                  "mlp.0.weight", "norm_slots.weight", "slot_mu", "predictor.blocks.0.attn.proj_o.weight"):
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
    close(mse, a["mse"], 1e-3, "mse")                                # the dVAE stays fp32 in both modes
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
