"""Times the fused trajectory-attention space step (fwd, bwd) at the bench shape.
usage: python tools/attn_bench.py [once]   ('once' = single fwd+bwd, for rocprofv3 --pmc)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
once = len(sys.argv) > 1
B, F, P, H, D = 8, 8, 196, 12, 64
S, C = F * P, H * D
torch.manual_seed(0)
qkv = (torch.randn(B, S + 1, 3 * C, device=dev) * 0.5).bfloat16().requires_grad_(True)


def run():
    xt, xd, cls = ops.traj_space(qkv, F, P, H)
    return xt, xd, cls


xt, xd, cls = run()
g = [torch.randn_like(xt), torch.randn_like(xd), torch.randn_like(cls)]
torch.autograd.backward([xt, xd, cls], g)
torch.cuda.synchronize()
if once:
    sys.exit(0)
n = 10
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(n):
    qkv.grad = None
    e[0].record()
    xt, xd, cls = run()
    e[1].record()
    torch.autograd.backward([xt, xd, cls], g)
    e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1])
    tb += e[1].elapsed_time(e[2])
fl = 4.0 * S * S * D * B * H
print("ATTN fwd %.1f us (%.0f TF/s)  bwd %.1f us (%.0f TF/s, 7 products)" % (
    tf / n * 1e3, fl / (tf / n) / 1e9, tb / n * 1e3, 3.5 * fl / (tb / n) / 1e9), flush=True)
