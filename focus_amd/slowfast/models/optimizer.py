"""Optimizer construction for the Motionformer path (mirror of slowfast/models/optimizer.py:48-172 for
OPTIMIZING_METHOD adamw/sgd: parameters named in model.no_weight_decay() and, with ZERO_WD_1D_PARAM, all 1-D
parameters get zero weight decay).  The update itself stays torch.optim (out of scope per SURVEY.md)."""
import torch


def construct_optimizer(model, cfg):
    skip = set()
    base = model.module if hasattr(model, "module") else model
    if hasattr(base, "no_weight_decay"):
        skip = base.no_weight_decay()
    decay, no_decay = [], []
    for name, p in base.named_parameters():
        if not p.requires_grad:
            continue
        if name in skip or (cfg.SOLVER.ZERO_WD_1D_PARAM and p.dim() == 1):
            no_decay.append(p)
        else:
            decay.append(p)
    groups = [{"params": decay, "weight_decay": cfg.SOLVER.WEIGHT_DECAY},
              {"params": no_decay, "weight_decay": 0.0}]
    assert len(decay) + len(no_decay) == len([p for p in base.parameters() if p.requires_grad])
    method = cfg.SOLVER.OPTIMIZING_METHOD
    if method == "adamw":
        fused = all(p.is_cuda for p in decay + no_decay)
        opt = torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-08, fused=fused)
    elif method == "sgd":
        opt = torch.optim.SGD(groups, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM,
                              dampening=cfg.SOLVER.DAMPENING, nesterov=cfg.SOLVER.NESTEROV)
    else:
        raise NotImplementedError("Does not support {} optimizer".format(method))
    # fused optimizers do not bump Tensor._version: tell the bf16 weight shadows that the masters moved
    from focus_amd import ops
    opt.register_step_post_hook(lambda *_a, **_k: ops.invalidate_shadows())
    return opt
