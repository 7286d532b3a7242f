"""Times the three HBM-streaming launches of the STEVE slot update directly through the C ABI (bf16, BASELINE configs[2]
shape by default: B=32, N=4096, D=192, K=11): focus_slot_attn_fwd, focus_slot_attn_bwd (deferred form: writes the
(w, dlogits) rows, forms dq) and focus_slot_kv_grad over 3 iterations.  Prints us per launch (main + finish kernels),
the algorithmic HBM bytes and the implied GB/s.  FOCUS_BENCH_LIB=<path to another build of libfocus_amd.so> times that build
instead (A/B against a previous commit).  usage: python tools/slot_kernel_bench.py [B N D K]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from focus_amd import _lib  # noqa: E402
from focus_amd.ops import _dt, _p, _stream  # noqa: E402

if os.environ.get("FOCUS_BENCH_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["FOCUS_BENCH_LIB"])


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B, N, D, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (32, 4096, 192, 11)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    bf = torch.bfloat16
    # several frames' worth of k/v so consecutive launches do not find their inputs in the 256 MB infinity cache
    NF = 6
    ks = [torch.randn(B, N, D, generator=g).to(dev, bf) for _ in range(NF)]
    vs = [torch.randn(B, N, D, generator=g).to(dev, bf) for _ in range(NF)]
    q = (torch.randn(B, K, D, generator=g) * D ** -0.5).to(dev, bf)
    dupd = torch.randn(B, K, D, generator=g).to(dev, bf)
    dattn = (torch.randn(B, N, K, generator=g) * 1e-3).to(dev, bf)
    attn = torch.empty(B, N, K, device=dev, dtype=bf)
    upd = torch.empty(B, K, D, device=dev, dtype=bf)
    cs = torch.empty(B, K, device=dev, dtype=torch.float32)
    dq = torch.empty_like(q)
    L = _lib.lib()
    nb = L.focus_slot_attn_workspace_bytes(B, N, K, D)
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    wls = [torch.empty(B, N, 32, device=dev, dtype=bf) for _ in range(3)]
    dk, dv = torch.empty_like(ks[0]), torch.empty_like(vs[0])
    it = [0]

    def fwd():
        i = it[0] = (it[0] + 1) % NF
        _lib.check(L.focus_slot_attn_fwd(_p(ks[i]), _p(vs[i]), N * D, _p(q), _p(attn), N * K, _p(upd), _p(cs), _p(ws), nb, B, N,
                                         K, D, 1e-8, _dt(q), _stream()), "fwd")

    def bwd():
        i = it[0] = (it[0] + 1) % NF
        _lib.check(L.focus_slot_attn_bwd(_p(ks[i]), _p(vs[i]), N * D, _p(q), _p(attn), N * K, _p(cs), _p(upd), _p(dupd),
                                         _p(dattn), None, None, 0, _p(dq), _p(ws), nb, B, N, K, D, 1e-8, _dt(q),
                                         _p(wls[i % 3]), _stream()), "bwd")

    def kvg():
        _lib.check(L.focus_slot_kv_grad(_p(wls[0]), _p(wls[1]), _p(wls[2]), None, _p(q), _p(q), _p(q), None, _p(dupd),
                                        _p(dupd), _p(dupd), None, 3, _p(dk), _p(dv), N * D, D, B, N, K, D, _dt(q), _stream()),
                   "kv_grad")

    fwd()
    kv = 2 * B * N * D * 2
    out = {"shape": [B, N, D, K]}
    for name, fn, nbytes in (("fwd", fwd, kv + B * N * K * 2), ("bwd_defer", bwd, kv + 2 * B * N * K * 2 + B * N * 64),
                             ("kv_grad_3it", kvg, kv + 3 * B * N * 64)):
        us = timed(fn)
        out[name] = {"us": round(us, 1), "MB": round(nbytes / 1e6, 1), "GB/s": round(nbytes / us / 1e3, 0)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
