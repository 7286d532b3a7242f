"""bf16 NT GEMM on the hot-path shapes: focus_amd's hand-written kernel next to the vendor library behind torch.matmul
(hipBLASLt / rocBLAS), same operands, back-to-back launches, HIP-event timed.  A measurement of headroom only: the
product path never calls the library.  usage (GPU box): python tools/gemm_vs_library.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from focus_amd import ops  # noqa: E402

SHAPES = [(100352, 768, 768), (12552, 3072, 768), (12552, 768, 3072), (12552, 2304, 768), (12552, 768, 768),
          (131072, 192, 192), (131072, 192, 384), (4096, 4096, 4096)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bt = b.t()
        t_own = timed(lambda: ops.mm_nt(a, b, out=c))
        t_lib = timed(lambda: torch.matmul(a, bt, out=c))
        fl = 2.0 * M * N * K
        print("%7d x %5d x %5d   own %7.1f us %5.0f TF/s   library %7.1f us %5.0f TF/s   own/library %.2f" % (
            M, N, K, t_own, fl / t_own / 1e6, t_lib, fl / t_lib / 1e6, t_lib / t_own), flush=True)


if __name__ == "__main__":
    main()
