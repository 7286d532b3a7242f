"""Hot-loop helpers and the job launcher (mirror of slowfast/utils/misc.py:26-33, 285-313, 388-398)."""
import math

import torch

from . import multiprocessing as mpu


def check_nan_losses(loss):
    if math.isnan(loss):
        raise RuntimeError("ERROR: Got NaN losses")


def launch_job(cfg, init_method, func, daemon=False):
    """misc.py:285-313: run `func(cfg)` on cfg.NUM_GPUS GPUs of this machine, one spawned process per GPU
    (mpu.run joins the process group and binds the device), or in-process when NUM_GPUS <= 1.
    The caller must not have initialised the GPU: the children are fresh interpreters ("spawn" start method) and a
    failing child makes spawn() raise in the parent, which then exits non-zero."""
    if cfg.NUM_GPUS > 1:
        torch.multiprocessing.spawn(
            mpu.run,
            nprocs=cfg.NUM_GPUS,
            args=(cfg.NUM_GPUS, func, init_method, cfg.SHARD_ID, cfg.NUM_SHARDS, cfg.DIST_BACKEND, cfg),
            daemon=daemon,
        )
    else:
        func(cfg=cfg)


def iter_to_cuda(batch):
    def _to(x):
        if isinstance(x, torch.Tensor):
            return x.cuda(non_blocking=True)
        if isinstance(x, (list, tuple)):
            return type(x)(_to(v) for v in x)
        if isinstance(x, dict):
            return {k: _to(v) for k, v in x.items()}
        return x
    return _to(batch)
