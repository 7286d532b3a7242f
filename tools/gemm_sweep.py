"""Times the bf16 MFMA GEMM on the hot-path shapes (random data) for each tuning variant.
usage (GPU box): python tools/gemm_sweep.py [4 8]  -- spawns one subprocess per FOCUS_GEMM_NLOAD value."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(100352, 768, 768), (12552, 3072, 768), (12552, 768, 3072), (12552, 2304, 768), (12552, 768, 768),
          (4096, 4096, 4096)]


def worker():
    sys.path.insert(0, ROOT)
    import torch
    from focus_amd import ops
    dev = torch.device("cuda:0")
    out = []
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        ref = None
        c = ops.mm_nt(a, b)
        if M <= 12552 and N <= 768:
            ref = (a[:64].float() @ b.float().t())
            err = float((c[:64].float() - ref).abs().max() / ref.abs().max())
        else:
            err = -1
        for _ in range(3):
            ops.mm_nt(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.mm_nt(a, b)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        out.append("%dx%dx%d: %.1f us %.0f TF/s err=%.1e" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
    print("loaders %s | " % os.environ.get("FOCUS_GEMM_NLOAD", "4") + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
    else:
        for v in (sys.argv[1:] or ["4", "8"]):          # loader waves per workgroup (FOCUS_GEMM_NLOAD)
            env = dict(os.environ, FOCUS_GEMM_NLOAD=v)
            subprocess.call([sys.executable, os.path.abspath(__file__), "worker"], env=env)
