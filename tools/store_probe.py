"""Does the GEMM epilogue's write rate depend on how contiguous a tile's rows are in C?  Same output bytes (77 MB),
K=64 (one K-step per tile, so the time is almost all epilogue): N=3072 (512-B row pieces at 6 KB stride) vs
N=256 (each 128x256 tile is one contiguous 64 KB block of C)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K) in [(12552, 3072, 64), (150624, 256, 64), (50208, 768, 64), (25104, 1536, 64)]:
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
    us = e0.elapsed_time(e1) * 50
    print("M=%d N=%d K=%d: %.1f us  -> C write %.2f TB/s" % (M, N, K, us, M * N * 2 / us / 1e6), flush=True)
