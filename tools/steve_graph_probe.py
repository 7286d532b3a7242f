"""Probe: the STEVE slot-update step (fwd+bwd) eager vs replayed from one HIP graph (torch.cuda.CUDAGraph)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo

B, T, N, D, K, IT = int(os.environ.get("B", 32)), 24, 4096, 192, 11, 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = SlotAttentionVideo(IT, K, D, D, 4 * D, num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, T, N, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
noise = torch.randn(B, K, D, device=dev, generator=g)
params = list(m.parameters())


def step():
    slots, attn = m(x, noise=noise)
    (slots.float().square().mean() + attn.float().mean()).backward()
    return slots


def timeit(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(2):
    step()
print("eager ms/step: %.2f" % timeit(step))
# capture
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        x.grad = None
        for p in params:
            p.grad = None
        step()
torch.cuda.current_stream().wait_stream(side)
x.grad = None
for p in params:
    p.grad = None
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph):
        out = step()
    print("graph ms/step: %.2f" % timeit(graph.replay))
    print("finite:", bool(torch.isfinite(out.float()).all()), "grad finite:", bool(torch.isfinite(x.grad.float()).all()))
except Exception as e:
    print("graph capture failed:", repr(e))
