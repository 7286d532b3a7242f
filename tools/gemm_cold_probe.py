"""Why does the NT GEMM run 25-40 % slower inside the step than in back-to-back launches?  Same kernel, same shapes,
HIP-event timed per launch, in three cache states:
  hot    : back-to-back launches of the same operands (tools/gemm_vs_library.py's regime)
  fresh  : the A operand is rewritten by another kernel (a copy) before every launch, as in the step, where the
           activation has just been produced by the preceding kernel
  flushed: a 512 MiB write to another buffer before every launch (cold L2 and Infinity Cache)
usage (GPU box): python tools/gemm_cold_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from focus_amd import ops  # noqa: E402

SHAPES = [(12552, 768, 768), (12552, 768, 3072), (12552, 2304, 768), (12552, 3072, 768), (12552, 768, 2304)]


def run(fn, pre, n=20):
    for _ in range(3):
        pre(); fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    print("shape                     hot     fresh-A   flushed   (us per launch, median of 20, events add ~4 us)")
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device=dev).bfloat16()
        src = a.clone()
        b = torch.randn(N, K, device=dev).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: ops.mm_nt(a, b, out=c)
        t_hot = run(fn, lambda: None)
        t_fresh = run(fn, lambda: a.copy_(src))
        t_cold = run(fn, lambda: big.fill_(1))
        # flushed, then ONE operand read back by another kernel (a sum): which operand's coldness costs?
        t_bcold = run(fn, lambda: (big.fill_(1), a.float().sum()))      # B (the weights) cold, A warm
        t_acold = run(fn, lambda: (big.fill_(1), b.float().sum()))      # A cold, B warm
        fl = 2.0 * M * N * K
        print("%6d x %5d x %5d  %7.1f  %7.1f  %7.1f   B-cold %7.1f  A-cold %7.1f   TF/s %5.0f %5.0f %5.0f" % (
            M, N, K, t_hot, t_fresh, t_cold, t_bcold, t_acold, fl / t_hot / 1e6, fl / t_fresh / 1e6, fl / t_cold / 1e6), flush=True)


if __name__ == "__main__":
    main()
