"""time(K) for a fixed M x N output: separates the per-K-step cost from the per-tile (epilogue) cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, N = 12552, 3072
res = []
for K in (64, 128, 256, 512, 768, 1536, 3072):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.mm_nt(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mm_nt(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    res.append("K=%d: %.1f us" % (K, e0.elapsed_time(e1) * 50))
print(os.environ.get("FOCUS_GEMM_WS", "-"), " | ".join(res), flush=True)
