"""Times ops.flash_attention (csrc/flash_attn.hip) at the STEVE decoder's shape: B*T sequences x 4 heads x 1024 tokens x 48.
usage: python tools/flash_bench.py [sequences=768] [p=0.1]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from focus_amd import ops  # noqa: E402

BT = int(sys.argv[1]) if len(sys.argv) > 1 else 768
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
H, N, D = 4, 1024, 48
dev = torch.device("cuda:0")
qkv = torch.randn(BT, N, 3 * H * D, device=dev).bfloat16().requires_grad_()
C = H * D
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
seed = torch.tensor([7], device=dev, dtype=torch.int32)
do = torch.randn(BT, N, C, device=dev).bfloat16()


def run(pp, back):
    out = ops.flash_attention(q, k, v, H, D ** -0.5, causal=True, p=pp, seed=seed)
    if back:
        qkv.grad = None
        out.backward(do)


for pp in (0.0, p):
    for back in (False, True):
        for _ in range(2):
            run(pp, back)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            run(pp, back)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        # algorithmic flops of the causal half: q.k and p.v forward (2 x 2 N^2/2 d); backward adds 5 such products
        fl = BT * H * (N * N / 2) * 2 * D * (2 + (5 if back else 0))
        print("p=%.2f %s: %.2f ms  %.0f TF/s (causal-half flops at d=48)" % (pp, "fwd+bwd" if back else "fwd", ms, fl / ms / 1e9))
