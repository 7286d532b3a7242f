"""One optimizer step of the supervised loop, in the reference's order (tools/train_net.py:68-121):
H2D -> forward -> loss -> NaN check -> zero_grad -> backward (DDP all-reduce overlapped) -> clip-norm -> step.

Differences that follow from the MI355X design and are deliberate:
  * TRAIN.MIXED_PRECISION selects bf16 storage with fp32 accumulation inside the HIP kernels instead of fp16
    autocast, so no GradScaler is needed (scale/unscale/update are identities);
  * the per-scalar blocking metric all-reduces (train_net.py:247-250) are packed into one collective
    (focus_amd/slowfast/utils/distributed.py:all_reduce).
"""
import torch

from .slowfast.utils import misc


def train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg, check_nan=True):
    preds = model(inputs, meta)                                   # train_net.py:86
    extra_preds = None
    if isinstance(preds, tuple):                                  # :87-88 (EK: (verb, {'verb','noun'}))
        preds, extra_preds = preds
    if cfg.TRAIN.DATASET == "epickitchens":                       # :95-97
        loss_dict = loss_fun(extra_preds, labels)
        loss = loss_dict["verb_loss"] + loss_dict["noun_loss"]
    else:
        loss = loss_fun(preds, labels)                            # :99
    if check_nan:
        misc.check_nan_losses(float(loss.detach()))                        # :102 (host sync, as in the reference)
    optimizer.zero_grad(set_to_none=True)                         # :105
    loss.backward()                                               # :106
    if cfg.SOLVER.CLIP_GRAD_VAL:
        torch.nn.utils.clip_grad_value_(model.parameters(), cfg.SOLVER.CLIP_GRAD_VAL)      # :108-111
    elif cfg.SOLVER.CLIP_GRAD_L2NORM and hasattr(optimizer, "step_clipped"):
        # :112-120 as one multi-tensor pass (optimizer.FusedAdamW: norm, clip, AdamW, bf16 shadows; csrc/optim.hip)
        optimizer.step_clipped(cfg.SOLVER.CLIP_GRAD_L2NORM)
        return preds, loss
    elif cfg.SOLVER.CLIP_GRAD_L2NORM:
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.CLIP_GRAD_L2NORM)    # :112-117
    optimizer.step()                                              # :120
    return preds, loss


def slot_train_step(model, optimizer, video, global_step, cfg, noise=None):
    """One iteration of slot_train_epoch (tools/steve_train_net.py:57-126): schedules -> forward -> mse + cross_entropy ->
    NaN check -> zero_grad -> backward -> clip-norm -> step.  Returns (loss, mse, cross_entropy, recon, attns, tau).
    TRAIN.MIXED_PRECISION selects bf16 storage inside the model (no autocast / GradScaler: see the module docstring)."""
    import math
    from .slowfast.models import optimizer as optim
    from .slowfast.utils import lr_policy as lrp
    so = cfg.SLOTS_OPTIM
    tau = lrp.cosine_anneal(global_step, so.TAU_START, so.TAU_FINAL, 0, so.TAU_STEPS)                # :60-66
    lr_warmup_factor_enc = lrp.linear_warmup(global_step, 0.0, 1.0, 0.0, so.WARMUP_STEPS)             # :68-73
    lr_warmup_factor_dec = lrp.linear_warmup(global_step, 0.0, 1.0, 0, so.WARMUP_STEPS)               # :75-80
    lr_decay_factor = math.exp(global_step / so.HALF_LIFE * math.log(0.5))                            # :82
    optim.set_slot_lr(optimizer, cfg, lr_decay_factor, lr_warmup_factor_enc, lr_warmup_factor_dec)    # :85-89
    recon, cross_entropy, mse, attns = model(video, tau, cfg.SLOTS.HARD) if noise is None else \
        model(video, tau, cfg.SLOTS.HARD, noise=noise)                                                # :98
    mse = mse.mean()                                                                                  # :101-102
    cross_entropy = cross_entropy.mean()
    loss = mse + cross_entropy                                                                        # :104
    misc.check_nan_losses(float(loss.detach()))                                                       # :108
    optimizer.zero_grad()                                                                             # :111
    loss.backward()
    if cfg.SOLVER.CLIP_GRAD_VAL:                                                                      # :116-123
        torch.nn.utils.clip_grad_value_(model.parameters(), cfg.SOLVER.CLIP_GRAD_VAL)
    elif cfg.SOLVER.CLIP_GRAD_L2NORM:
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.SOLVER.CLIP_GRAD_L2NORM)
    optimizer.step()                                                                                  # :126
    return loss, mse, cross_entropy, recon, attns, tau


def slot_train_epoch(train_loader, model, optimizer, scaler, train_meter, cur_epoch, cfg, writer=None):
    """tools/steve_train_net.py:33-160 with the reference's argument list (scaler and train_meter are accepted and unused:
    bf16 needs no loss scaling, and the reference never touches the meter either).  `train_loader` yields video tensors
    [B,T,C,H,W]; `writer`, if given, receives the same scalar dict through add_scalars(dict, global_step=...).
    Returns {'tau', 'global_step'} like the reference (:153-158).  The end-of-epoch autoregressive visualisation
    (:146-150, slot_misc.visualize + add_video) is dataset/tensorboard plumbing and is not part of this path."""
    model.train()
    data_size = len(train_loader)
    tau, global_step = None, cur_epoch * data_size
    for cur_iter, video in enumerate(train_loader):
        global_step = cur_epoch * data_size + cur_iter
        if cfg.NUM_GPUS:
            video = video.cuda(non_blocking=True)
        loss, mse, ce, _recon, _attns, tau = slot_train_step(model, optimizer, video, global_step, cfg)
        if writer is not None:
            with torch.no_grad():
                writer.add_scalars({"TRAIN/loss": loss.item(), "TRAIN/cross_entropy": ce.item(), "TRAIN/mse": mse.item(),
                                    "TRAIN/tau": tau, "TRAIN/lr_dvae": optimizer.param_groups[0]["lr"],
                                    "TRAIN/lr_enc": optimizer.param_groups[1]["lr"],
                                    "TRAIN/lr_dec": optimizer.param_groups[2]["lr"]}, global_step=global_step)
    return {"tau": tau, "global_step": global_step}


@torch.no_grad()
def slot_eval_epoch(eval_loader, model, cfg=None):
    """FG-ARI evaluation of STEVE (tools/steve_eval_net.py:75-132): `eval_loader` yields (video [B,T,C,H,W], true_masks
    [B,T,S,1,H,W] with segment 0 = background); the slots' attention masks of `model.encode` are scored against the
    foreground segments over the whole clip (pixels of all frames flattened together).  Returns (mean, std over batches)
    of 100 x ARI; the ARI itself is computed on the device (utils/metrics.evaluate_ari)."""
    from .slowfast.utils import metrics
    model.eval()
    dev = next(model.parameters()).device
    scores = []
    for video, true_masks in eval_loader:
        video = video.to(dev)
        _, _, pred_masks = model.encode(video)                                            # [B,T,K,1,H,W]
        scores.append(100 * metrics.evaluate_ari(true_masks.to(dev).permute(0, 2, 1, 3, 4, 5)[:, 1:].flatten(start_dim=2),
                                                 pred_masks.permute(0, 2, 1, 3, 4, 5).flatten(start_dim=2)))
    t = torch.tensor(scores, dtype=torch.float64)
    return float(t.mean()), float(t.std(unbiased=False)) if len(scores) > 1 else 0.0


@torch.no_grad()
def perform_test(test_loader, model, test_meter, cfg):
    """Multi-view testing (tools/test_net.py:24-157, single-label branch): every clip of every video goes through the
    model in eval mode (softmax scores), the meter sums the views per video.  `test_loader` yields
    (inputs, labels, video_idx, meta) like the reference's loaders."""
    model.eval()
    dev = next(model.parameters()).device
    for inputs, labels, video_idx, meta in test_loader:
        inputs = [x.to(dev) for x in inputs] if isinstance(inputs, (list, tuple)) else inputs.to(dev)
        meta = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in meta.items()}
        preds = model(inputs, meta)
        if isinstance(preds, tuple):
            preds = preds[0]
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            from .slowfast.utils import distributed as du
            preds, labels, video_idx = du.all_gather([preds, labels.to(dev), video_idx.to(dev)])
        test_meter.update_stats(preds, labels, video_idx)
    return test_meter.finalize_metrics()


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


def live_autograd_tensors(device=None):
    """CUDA tensors that still hang on an autograd graph (grad_fn is not None), found through the garbage collector's
    object list: the outputs of an earlier forward that somebody still references."""
    import gc
    import warnings
    found = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # isinstance() on lazy module proxies (torch.distributed.reduce_op) warns
        for o in gc.get_objects():
            try:
                if isinstance(o, torch.Tensor) and o.is_cuda and o.grad_fn is not None and (device is None or o.device == device):
                    found.append(o)
            except Exception:       # objects in odd states while being torn down
                pass
    return found


class GraphedStep:
    """One HIP graph for a launch-bound step (the STEVE slot update issues ~2000 kernels of 3-30 us per step: the host,
    not the GPU, sets its eager time).  `fn` takes no arguments, reads its inputs from tensors that stay in place, and
    runs forward + backward (gradients it leaves in .grad are rewritten by every replay).  The class owns the warm-up
    runs (on a side stream, as torch.cuda.graphs requires) and drops their outputs; `replay()` re-issues the whole
    step with one launch.

    Failure this class guards against (gpurun_out/r2_b42.log: `Segmentation fault in torch/cuda/graphs.py capture_end`
    on ROCm 7.2): an output of an EARLIER run of the step that is still referenced keeps that run's autograd graph --
    and the buffers it saved -- alive across the capture, and ending the capture then crashes inside the HIP runtime.
    The cause sits with whoever holds the reference, so the guard sits here, before the capture starts: after the
    warm-up the collector runs, the device is synchronised, and if ANY CUDA tensor with a grad_fn is still reachable
    (the caller's outputs of an earlier eager step, or outputs `fn` stores somewhere itself) the constructor raises
    instead of capturing.  `allow_live` names tensors the caller knows about and vouches for."""

    def __init__(self, fn, warmup=2, reset=None, allow_live=()):
        """reset: called before every warm-up run and before the capture (e.g. set .grad to None, so that the captured
        backward ASSIGNS gradients instead of accumulating into tensors from outside the graph)."""
        import gc
        import weakref
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        refs = []
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if reset is not None:
                    reset()
                out = fn()
                refs.extend(weakref.ref(t) for t in _tensors(out))
                del out
        cur.wait_stream(side)
        if reset is not None:
            reset()
        gc.collect()
        torch.cuda.synchronize()
        if any(r() is not None for r in refs):
            raise RuntimeError("GraphedStep: fn keeps its own outputs alive between calls (a warm-up output is still referenced "
                               "after `del`); capturing now would crash in hipStreamEndCapture")
        ok = {id(t) for t in allow_live}
        live = [t for t in live_autograd_tensors(torch.device("cuda", torch.cuda.current_device())) if id(t) not in ok]
        if live:
            def who(t):
                try:
                    return [type(r).__name__ + (":" + ",".join(k for k, v in r.items() if v is t)[:60] if isinstance(r, dict) else "")
                            for r in gc.get_referrers(t) if r is not live][:3]
                except Exception:
                    return []
            raise RuntimeError("GraphedStep: %d CUDA tensor(s) of an earlier forward are still referenced (shape, grad_fn, held by: "
                               "%s); their autograd graph would be alive across the capture (crash in hipStreamEndCapture on "
                               "ROCm 7.2).  Drop them (del / detach) before building the graph" %
                               (len(live), [(tuple(t.shape), type(t.grad_fn).__name__, who(t)) for t in live[:4]]))
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs


def synthetic_batch(cfg, batch, device, seed=0):
    """Synthetic clips of BASELINE.md section 3: frames ~ N(0,1) [B,3,T,H,W], boxes [B,T,O,4] cxcywh with
    cx,cy ~ U(0.3,0.7), w,h ~ U(0.1,0.5) kept inside the frame, one object slot emptied, labels ~ randint.
    Drawn with a CPU generator so every box/rank sees the same numbers for a given seed."""
    g = torch.Generator().manual_seed(seed)
    Tn, S, O = cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE, cfg.ORVIT.O
    x = torch.randn(batch, 3, Tn, S, S, generator=g)
    wh = 0.1 + 0.4 * torch.rand(batch, Tn, O, 2, generator=g)
    c = 0.3 + 0.4 * torch.rand(batch, Tn, O, 2, generator=g)
    c = torch.minimum(torch.maximum(c, wh / 2), 1 - wh / 2)
    boxes = torch.cat([c, wh], dim=-1)
    boxes[batch - 1, :, O - 1] = 0                                 # empty-box path
    if cfg.TRAIN.DATASET == "epickitchens":                        # verb (97) / noun (300) label dict (train_net.py:95)
        labels = {"verb": torch.randint(0, 97, (batch,), generator=g).to(device),
                  "noun": torch.randint(0, 300, (batch,), generator=g).to(device)}
    else:
        labels = torch.randint(0, cfg.MODEL.NUM_CLASSES, (batch,), generator=g).to(device)
    return [x.to(device)], labels, {"orvit_bboxes": boxes.to(device)}
