"""The fused optimizer step (csrc/optim.hip through optimizer.FusedAdamW) against torch's own clip_grad_norm_ + AdamW
(the sequence of tools/train_net.py:112-120) on identical parameters and gradients: fp32 arithmetic, tolerance 1e-6
relative per tensor over 3 steps; the bf16 shadows it writes must equal the rounded masters bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


SHAPES = [(768, 768), (2304, 768), (768, 3072), (174, 768), (768,), (3,), (1, 197, 768), (768, 3, 2, 16, 16), (8, 4, 768), (6, 10),
          (4100,), (64, 68)]


def _params(seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.05).to(dev())) for s in SHAPES]


@pytest.mark.parametrize("max_norm", [0.05, 1e9, 0.0])
def test_fused_adamw_matches_torch(max_norm):
    from focus_amd import ops
    from focus_amd.slowfast.models.optimizer import FusedAdamW
    ops.drop_caches()
    pa, pb = _params(0), _params(0)
    groups = lambda ps: [{"params": [p for p in ps if p.dim() > 1], "weight_decay": 5e-2},
                         {"params": [p for p in ps if p.dim() <= 1], "weight_decay": 0.0}]
    ref = torch.optim.AdamW(groups(pa), lr=3e-3, eps=1e-8)
    fus = FusedAdamW(groups(pb), lr=3e-3, eps=1e-8)
    # two of the weights have live bf16 shadows (row-major and transposed), as after a forward/backward
    sh = [(pb[0], ops.shadow(pb[0], torch.bfloat16), ops.shadow(pb[0], torch.bfloat16, transposed=True)),
          (pb[1], ops.shadow(pb[1], torch.bfloat16), None)]
    g = torch.Generator(device="cpu").manual_seed(7)
    for step in range(3):
        grads = [torch.randn(*s, generator=g).to(dev()) * (0.3 if step != 1 else 1e-4) for s in SHAPES]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if step == 2:                                # a parameter without gradient is skipped by both
            pa[3].grad = pb[3].grad = None
        if max_norm > 0:
            tn = torch.nn.utils.clip_grad_norm_(pa, max_norm)
        ref.step()
        fus.step_clipped(max_norm)
        if max_norm > 0:
            assert abs(float(fus.last_total_norm) - float(tn)) <= 2e-6 * float(tn)
        for i, (p, q) in enumerate(zip(pa, pb)):
            d = float((p.detach() - q.detach()).abs().max())
            assert d <= 1e-6 * max(float(p.detach().abs().max()), 1e-3), "step %d tensor %d: %.3e" % (step, i, d)
            if p.grad is not None:                   # clip_grad_norm_ leaves the clipped gradient in .grad
                dg = float((p.grad - q.grad).abs().max())
                assert dg <= 2e-6 * max(float(p.grad.abs().max()), 1e-12), "step %d grad %d: %.3e" % (step, i, dg)
        for w, d, dT in sh:
            assert torch.equal(d, w.detach().to(torch.bfloat16)), "row-major shadow is not the rounded master"
            assert ops.shadow(w, torch.bfloat16) is d, "the shadow written by the step must be the fresh one"
            if dT is not None:
                assert torch.equal(dT, w.detach().to(torch.bfloat16).t().contiguous())
                assert ops.shadow(w, torch.bfloat16, transposed=True) is dT
    # state_dict layout is torch.optim.AdamW's
    sa, sb = ref.state_dict(), fus.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys()
    for k in sa["state"]:
        assert set(sa["state"][k].keys()) == set(sb["state"][k].keys())
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    # round trip through state_dict keeps stepping
    fus2 = FusedAdamW(groups(pb), lr=3e-3, eps=1e-8)
    fus2.load_state_dict(sb)
    for q in pb:
        q.grad = torch.ones_like(q)
    before = [q.detach().clone() for q in pb]
    fus2.step()
    assert all(float((b - q.detach()).abs().max()) > 0 for b, q in zip(before, pb))
    assert float(fus2.state[pb[0]]["step"]) == 4.0
