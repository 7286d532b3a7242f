"""Times the TN (weight-gradient) MFMA GEMM on hot-path shapes.  usage: python tools/tn_sweep.py [once]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
once = len(sys.argv) > 1
out = []
for (N, K, M) in [(3072, 768, 12552), (768, 3072, 12552), (768, 768, 12552), (2304, 768, 12552), (768, 768, 100352), (384, 768, 50176), (768, 1536, 12544)]:
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    c = ops.mm_tn(dy, x)
    if once:
        torch.cuda.synchronize()
        continue
    ref = dy[:, :32].float().t() @ x.float()
    err = float((c[:32] - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        ops.mm_tn(dy, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.mm_tn(dy, x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    out.append("%dx%dx%d: %.1f us %.0f TF/s err=%.1e" % (N, K, M, us, 2.0 * M * N * K / us / 1e6, err))
print("TN | " + " | ".join(out), flush=True)
