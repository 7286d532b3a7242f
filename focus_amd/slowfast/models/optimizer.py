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


def construct_optimizer_slot(model, cfg):
    """optimizer.py:13-40: three parameter groups -- dVAE, encoder, decoder -- whose learning rates set_slot_lr rewrites
    every step."""
    base = model.module if hasattr(model, "module") else model
    named = list(base.named_parameters())
    optim_params = [
        {"params": [p for n, p in named if "dvae" in n], "lr": cfg.SLOTS_OPTIM.DVAE},
        {"params": [p for n, p in named if "steve_encoder" in n], "lr": 0.0},
        {"params": [p for n, p in named if "steve_decoder" in n], "lr": 0.0},
    ]
    method = cfg.SOLVER.OPTIMIZING_METHOD
    if method == "sgd":
        opt = torch.optim.SGD(optim_params, lr=cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM,
                              weight_decay=cfg.SOLVER.WEIGHT_DECAY, dampening=cfg.SOLVER.DAMPENING,
                              nesterov=cfg.SOLVER.NESTEROV)
    elif method == "adam":
        opt = torch.optim.Adam(optim_params)
    else:
        raise NotImplementedError("Does not support {} optimizer".format(method))
    from focus_amd import ops
    opt.register_step_post_hook(lambda *_a, **_k: ops.invalidate_shadows())
    return opt


def set_slot_lr(optimizer, cfg, lr_decay_factor, lr_warmup_factor_enc, lr_warmup_factor_dec):
    """optimizer.py:213-222."""
    optimizer.param_groups[0]["lr"] = cfg.SLOTS_OPTIM.DVAE
    optimizer.param_groups[1]["lr"] = lr_decay_factor * lr_warmup_factor_enc * cfg.SLOTS_OPTIM.ENC
    optimizer.param_groups[2]["lr"] = lr_decay_factor * lr_warmup_factor_dec * cfg.SLOTS_OPTIM.DEC
