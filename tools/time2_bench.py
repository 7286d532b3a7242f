"""Times the temporal step at the bench shape (B=8, S=1568, F=8, 12 heads): k2 path vs k2-free path, forward and
backward separately (HIP events, median of 7 rounds, interleaved)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focus_amd import ops

B, S, F_, heads = int(os.environ.get("B", 8)), 1568, 8, 12
C = heads * 64
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
q2 = torch.randn(B, S, C, device=dev, generator=g).bfloat16().requires_grad_()
xt = torch.randn(B, S, F_, C, device=dev, generator=g).bfloat16().requires_grad_()
w = (torch.randn(2 * C, C, device=dev, generator=g) * C ** -0.5).requires_grad_()
b = torch.zeros(2 * C, device=dev).requires_grad_()
cls = torch.randn(B, 1, C, device=dev, generator=g).bfloat16()
ct = torch.randn(B, S + 1, C, device=dev, generator=g).bfloat16()
fns = {"k2": ops.traj_time_block, "k2-free": ops.traj_time2_block}
res = {k: {"fwd": [], "bwd": []} for k in fns}
for rnd in range(9):
    for name, fn in fns.items():
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        out = fn(q2, xt, w, b, cls, heads)
        e[1].record()
        out.backward(ct)
        e[2].record()
        torch.cuda.synchronize()
        q2.grad = xt.grad = w.grad = b.grad = None
        if rnd >= 2:
            res[name]["fwd"].append(e[0].elapsed_time(e[1]) * 1e3)
            res[name]["bwd"].append(e[1].elapsed_time(e[2]) * 1e3)
for name in fns:
    print("%-8s fwd %7.1f us   bwd %7.1f us" % (name, statistics.median(res[name]["fwd"]), statistics.median(res[name]["bwd"])))
