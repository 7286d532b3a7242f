"""Optimizer construction for the Motionformer path (mirror of slowfast/models/optimizer.py:48-172 for
OPTIMIZING_METHOD adamw/sgd: parameters named in model.no_weight_decay() and, with ZERO_WD_1D_PARAM, all 1-D
parameters get zero weight decay).  The update itself stays torch.optim (out of scope per SURVEY.md)."""
import os

import torch


class FusedAdamW(torch.optim.AdamW):
    """torch.optim.AdamW (same param_groups / state_dict layout: state['step'], ['exp_avg'], ['exp_avg_sq']) whose step
    is focus_adamw_step: gradient-norm clipping (train_net.py:112-117), the AdamW update and the bf16 weight shadows of
    focus_amd.ops in two launches over all parameters.  `step_clipped(max_norm)` is what focus_amd.train.train_step calls
    in place of clip_grad_norm_ + step(); plain `step()` is the same without clipping.  `last_total_norm` is the device
    scalar clip_grad_norm_ would have returned."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=False, foreach=False)
        self._table = None          # dict: the device tables of the last step (see _prepare)
        self._groups = None         # (values, device tensor)
        self._ws = None
        self.last_total_norm = None
        self.write_clipped_grads = True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = None

    def _prepare(self):
        import numpy as np
        from focus_amd import _lib, ops
        L = _lib.lib()
        live = []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdamW: amsgrad / maximize are not built")
            for p in group["params"]:
                if p.grad is not None:
                    live.append((p, gi))
        if not live:
            return None
        dev = live[0][0].device
        ident = (tuple(id(p) for p, _ in live), ops.shadow_epoch())
        tab = self._table
        if tab is None or tab["ident"] != ident:
            # parameters, moments, step counters and shadows: these pointers only change when the set of live
            # parameters, the optimizer state (load_state_dict) or the shadow cache (first forward) does
            n = len(live)
            for p, _ in live:
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("FusedAdamW wants dense contiguous fp32 parameters on the GPU")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps = torch.stack([self.state[p]["step"].to(dev, torch.float32).reshape(()) for p, _ in live])
            rec = np.zeros((n, 8), dtype=np.int64)
            unit, entries = 0, []
            for i, (p, gi) in enumerate(live):
                st = self.state[p]
                st["step"] = steps[i]                                  # 0-dim view: state_dict() still sees one tensor per param
                rows, cols = (p.shape[0], p.numel() // p.shape[0]) if p.dim() >= 2 else (1, p.numel())
                tile = int(p.dim() == 2 and rows % 4 == 0 and cols % 4 == 0 and p.data_ptr() % 16 == 0
                           and st["exp_avg"].data_ptr() % 16 == 0 and st["exp_avg_sq"].data_ptr() % 16 == 0)
                dst = ops.cached_shadow(p, torch.bfloat16, False) if p.dim() == 2 else None
                dstT = ops.cached_shadow(p, torch.bfloat16, True) if tile else None
                rec[i, 0], rec[i, 1], rec[i, 2] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                rec[i, 3] = dst.data_ptr() if dst is not None else 0
                rec[i, 4] = dstT.data_ptr() if dstT is not None else 0
                rec[i, 5] = int(rows) | (int(cols) << 32)
                rec[i, 6] = int(gi) | (int(unit) << 32)
                rec[i, 7] = tile
                entries.append((p, dst, dstT))
                unit += L.focus_adamw_units(int(rows), int(cols), tile)
            tab = self._table = {"ident": ident, "items": torch.from_numpy(rec).pin_memory().to(dev, non_blocking=True),
                                 "units": unit, "steps": steps, "entries": entries, "tile": rec[:, 7].copy(),
                                 "gsig": None, "gptrs": None}
        grads = []
        for i, (p, _) in enumerate(live):
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or (tab["tile"][i] and g.data_ptr() % 16):
                p.grad = g = g.float().contiguous().clone() if tab["tile"][i] and g.data_ptr() % 16 else g.float().contiguous()
            grads.append(g.data_ptr())
        gsig = tuple(grads)
        if tab["gsig"] != gsig:
            tab["gsig"] = gsig
            tab["gptrs"] = torch.tensor(grads, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        vals = tuple((float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups)
        if self._groups is None or self._groups[0] != vals or self._groups[1].device != dev:
            self._groups = (vals, torch.tensor(vals, dtype=torch.float32).reshape(-1, 2).pin_memory().to(dev, non_blocking=True))
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(L.focus_adamw_workspace_bytes() // 4 + 1, dtype=torch.float32, device=dev)
        return dev

    @torch.no_grad()
    def step_clipped(self, max_norm=0.0, closure=None):
        import ctypes
        from focus_amd import _lib, ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        dev = self._prepare()
        if dev is None:
            return loss
        tab = self._table
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for g in self.param_groups[1:]:
            if tuple(g["betas"]) != (b1, b2) or g["eps"] != eps:
                raise NotImplementedError("FusedAdamW: one (betas, eps) for all groups")
        L = _lib.lib()
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        norm = self._ws[-1:]
        with torch.cuda.device(dev):
            _lib.check(L.focus_adamw_step(vp(tab["items"]), vp(tab["gptrs"]), len(tab["entries"]), tab["units"],
                                          vp(self._groups[1]), vp(tab["steps"]), vp(self._ws), (self._ws.numel() - 1) * 4,
                                          vp(norm), b1, b2, eps, float(max_norm or 0.0), int(self.write_clipped_grads),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "adamw_step")
        self.last_total_norm = norm[0]
        ops.shadows_written(tab["entries"])
        return loss

    def step(self, closure=None):
        return self.step_clipped(0.0, closure)


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
        on_gpu = all(p.is_cuda and p.dtype == torch.float32 for p in decay + no_decay)
        if on_gpu and os.environ.get("FOCUS_FUSED_OPT", "1") != "0":
            # clip + AdamW + bf16 shadows in two launches (csrc/optim.hip); no post-step hook: the step writes the shadows
            return FusedAdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-08)
        opt = torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, eps=1e-08, fused=on_gpu)
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
