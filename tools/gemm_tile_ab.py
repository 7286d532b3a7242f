"""A/B of the NT GEMM row-tile height (128x256 three-stage vs 192x256 two-stage) on the bench's shapes, interleaved
rounds in one process (cdna_hip_programming.md rule 24), random bf16 operands.  Prints median us and TF/s."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from focus_amd import _lib, ops

SHAPES = [(12552, 768, 768), (12552, 768, 2304), (12552, 768, 3072), (12552, 2304, 768), (12552, 3072, 768),
          (100352, 768, 768), (12808, 768, 768), (12808, 2304, 768), (6272, 768, 768), (28224, 768, 768),
          (28224, 2304, 768), (28224, 3072, 768), (28224, 768, 3072)]
dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {128: [], 192: []}
    ref = None
    for bm in (128, 192):
        L.focus_gemm_tile_override(bm)
        for _ in range(3):
            ops.mm_nt(a, w, out=out)
        if ref is None:
            ref = out.clone()
        else:
            assert torch.equal(ref, out), "tile variants disagree for %s" % ((M, N, K),)
    for rnd in range(7):
        for bm in (128, 192):
            L.focus_gemm_tile_override(bm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.mm_nt(a, w, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[bm].append(e0.elapsed_time(e1) * 100.0)          # us per call
    L.focus_gemm_tile_override(0)
    fl = 2.0 * M * N * K
    m128, m192 = statistics.median(res[128]), statistics.median(res[192])
    print("%-22s 128x256: %7.1f us %6.0f TF/s | 192x256: %7.1f us %6.0f TF/s | ratio %.2f" %
          ((M, N, K), m128, fl / m128 / 1e6, m192, fl / m192 / 1e6, m128 / m192))
