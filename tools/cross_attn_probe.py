"""Times the STEVE decoder's cross-attention (1024 image tokens -> 11 slots, 4 heads x 48) through ops.flash_attention and its
self-attention at the same batch:   python tools/cross_attn_probe.py [sequences=192] [p=0.1]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from focus_amd import ops  # noqa: E402

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 192
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
H, N, D, S = 4, 1024, 48, 11
C = H * D
dev = torch.device("cuda:0")
seed = torch.tensor([7], device=dev, dtype=torch.int32)
for name, nk, causal in (("cross 1024 -> 11", S, False), ("self 1024 causal", N, True)):
    q = torch.randn(BT, N, C, device=dev).bfloat16().requires_grad_()
    kv = torch.randn(BT, nk, 2 * C, device=dev).bfloat16().requires_grad_()
    k, v = kv[..., :C], kv[..., C:]
    do = torch.randn(BT, N, C, device=dev).bfloat16()
    assert ops.flash_ok(q, k, v, H, causal)
    for back in (False, True):
        def run():
            out = ops.flash_attention(q, k, v, H, D ** -0.5, causal=causal, p=p, seed=seed)
            if back:
                q.grad = kv.grad = None
                out.backward(do)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        print("%-18s %-8s %.3f ms" % (name, "fwd+bwd" if back else "fwd", 1e3 * (time.perf_counter() - t0) / 5), flush=True)
