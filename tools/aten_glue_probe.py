"""Which Python lines launch the ATen glue kernels of one bench step?  torch.profiler with stacks, grouped by
(kernel-launching aten op, innermost focus_amd/bench source line)."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from focus_amd.slowfast.models import build_model
from focus_amd.slowfast.models.losses import get_loss_func
from focus_amd.slowfast.models.optimizer import construct_optimizer
from focus_amd.train import synthetic_batch, train_step
from torch.profiler import profile, ProfilerActivity

cfg = bench.make_cfg(1, 8)
torch.manual_seed(0)
torch.cuda.set_device(0)
m = build_model(cfg); m.train()
opt = construct_optimizer(m, cfg)
lf = get_loss_func(cfg)(reduction="mean")
inputs, labels, meta = synthetic_batch(cfg, 8, "cuda", seed=1)
for _ in range(3):
    train_step(m, opt, lf, inputs, labels, meta, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(m, opt, lf, inputs, labels, meta, cfg)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.self_device_time_total > 0:
        src = "?"
        for fr in (ev.stack or []):
            if "focus_amd" in fr or "bench.py" in fr or "train.py" in fr:
                src = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-70:]
                break
        k = (ev.name, src)
        agg[k][0] += 1
        agg[k][1] += ev.self_device_time_total
tot = sum(v[1] for v in agg.values())
print("aten ops with device time: %.3f ms/step" % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.1f us %4d calls  %-28s %s" % (v[1], v[0], k[0], k[1]))
