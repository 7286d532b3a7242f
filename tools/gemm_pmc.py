"""Runs a few MFMA GEMM shapes once each (for rocprofv3 --pmc passes).  usage: python tools/gemm_pmc.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K) in [(12552, 3072, 768), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        ops.mm_nt(a, b)
    torch.cuda.synchronize()
