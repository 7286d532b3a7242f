"""Which source lines issue the ATen kernels of one bench step?  A TorchDispatchMode records every non-view aten op
with the innermost focus_amd/ frame of the Python stack (forward and custom-Function backward alike) and the bytes it
touches; grouped by (op, line)."""
import collections, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from focus_amd.slowfast.models import build_model
from focus_amd.slowfast.models.losses import get_loss_func
from focus_amd.slowfast.models.optimizer import construct_optimizer
from focus_amd.train import synthetic_batch, train_step
from torch.utils._python_dispatch import TorchDispatchMode

VIEWS = {"view", "reshape", "_unsafe_view", "expand", "permute", "transpose", "t", "slice", "select", "unsqueeze", "squeeze",
         "as_strided", "detach", "alias", "unbind", "split", "split_with_sizes", "narrow", "flatten", "unflatten", "chunk",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "_reshape_alias", "lift_fresh", "is_same_size",
         "sym_size", "sym_stride", "sym_numel", "stride", "size", "numel", "dim", "is_contiguous", "_local_scalar_dense",
         "set_", "resize_", "view_as_real", "result_type", "item"}
agg = collections.defaultdict(lambda: [0, 0])


class Probe(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            src = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/focus_amd/" in fr.filename or fr.filename.endswith("train.py"):
                    src = "%s:%d" % (fr.filename.split("/root/repo/")[-1].split("focus_amd/")[-1], fr.lineno)
                    break
            nbytes = 0
            for o in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(o, torch.Tensor):
                    nbytes += o.numel() * o.element_size()
            k = (name, src)
            agg[k][0] += 1
            agg[k][1] += nbytes
        return out


cfg = bench.make_cfg(1, 8)
torch.manual_seed(0)
torch.cuda.set_device(0)
m = build_model(cfg); m.train()
opt = construct_optimizer(m, cfg)
lf = get_loss_func(cfg)(reduction="mean")
inputs, labels, meta = synthetic_batch(cfg, 8, "cuda", seed=1)
for _ in range(2):
    train_step(m, opt, lf, inputs, labels, meta, cfg)
torch.cuda.synchronize()
with Probe():
    train_step(m, opt, lf, inputs, labels, meta, cfg)
torch.cuda.synchronize()
print("%d aten ops" % sum(v[0] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print("%5d calls %9.2f MB out  %-26s %s" % (v[0], v[1] / 1e6, k[0], k[1]))
