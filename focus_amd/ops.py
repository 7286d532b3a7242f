"""Autograd operators over the C ABI of libfocus_amd.so.

Every function here enqueues hand-written HIP kernels on torch's current stream through ctypes;
PyTorch only owns memory and the autograd tape.  CPU tensors are rejected: the hot path has no
fallback (the CPU restatement lives in oracle/ and is test infrastructure only).
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import BF16, F32, GemmDesc

EPI_NONE, EPI_GELU, EPI_RELU, EPI_TANH = _lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RELU, _lib.EPI_TANH
_DEPI = {EPI_GELU: _lib.EPI_DGELU, EPI_RELU: _lib.EPI_DRELU, EPI_TANH: _lib.EPI_DTANH}


# --------------------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------------------
def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError("focus_amd: unsupported dtype %s (float32 or bfloat16)" % t.dtype)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("focus_amd: the hot path runs on the MI355X only (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, off=0):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


USE_TN_GEMM = True   # weight gradients straight from row-major activations (gemm_mfma_tn.hip)

# bench.py sets this to a list to collect (flops, start_event, end_event) for every MFMA-path GEMM launch
GEMM_TIMING = None


def _mfma_path(ta, sA, sB, K):
    """'nt' / 'tn' when the C dispatcher will take that MFMA kernel for this descriptor, else None."""
    if ta.dtype != torch.bfloat16:
        return None
    if sA[1] == 1 and sB[0] == 1 and K > 0 and K % 64 == 0 and sA[0] % 8 == 0 and sB[1] % 16 == 0:
        return "nt"
    if sA[1] == 1 and sB[0] == 1 and K > 0 and K % 64 == 0 and sA[0] % 8 == 0 and sB[1] % 8 == 0:
        return "nt"
    if sA[0] == 1 and sB[1] == 1 and sA[1] % 8 == 0 and sB[0] % 8 == 0:
        return "tn"
    return None


def gemm(M, N, K, A, sA, B, sB, C, sC, batch=(1, 1), bias=None, residual=None, aux=None, alpha=1.0,
         accumulate=False, epilogue=EPI_NONE, b_scale=None):
    """C = epi(alpha*A.B + bias) + residual.  A/B/C are (tensor, element_offset); s* = (rs, cs, bs0, bs1).
    b_scale: B holds OCP e4m3 codes (a uint8 tensor) with this per-tensor fp32 scale (device scalar); A is bf16."""
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.batch0, d.batch1 = batch
    (ta, oa), (tb, ob), (tc, oc) = A, B, C
    d.A, d.B, d.C = _p(ta, oa).value, _p(tb, ob).value, _p(tc, oc).value
    d.rsA, d.csA, d.bsA0, d.bsA1 = sA
    d.rsB, d.csB, d.bsB0, d.bsB1 = sB
    d.rsC, d.csC, d.bsC0, d.bsC1 = sC
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual = (residual[0].data_ptr() + residual[1] * residual[0].element_size()) if residual is not None else None
    d.aux = (aux[0].data_ptr() + aux[1] * aux[0].element_size()) if aux is not None else None
    d.alpha = alpha
    d.accumulate = int(accumulate)
    d.epilogue = epilogue
    d.dtype_ab = _dt(ta)
    d.dtype_c = _dt(tc)
    if b_scale is not None:
        assert tb.dtype == torch.uint8 and ta.dtype == torch.bfloat16, "fp8 weights go with bf16 activations"
        d.dtype_b = _lib.FP8_E4M3
        d.b_scale = b_scale.data_ptr()
    else:
        assert ta.dtype == tb.dtype
    kind = _mfma_path(ta, sA, sB, K) if GEMM_TIMING is not None else None
    if kind:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().focus_gemm(ctypes.byref(d), _stream()), "gemm")
        e1.record()
        if kind == "nt" and _lib.lib().focus_gemm_last_kernel() == 2:
            kind = "nt_ws8" if b_scale is not None else "nt_ws"   # the wave-specialised kernel (gemm_mfma_ws.hip)
        GEMM_TIMING.append((2.0 * M * N * K * batch[0] * batch[1], e0, e1, kind, (M, N, K, batch[0] * batch[1], epilogue)))
        return
    _lib.check(_lib.lib().focus_gemm(ctypes.byref(d), _stream()), "gemm")


def mm_nt(a, b, bias=None, residual=None, aux=None, epilogue=EPI_NONE, out_dtype=None, alpha=1.0, out=None, b_scale=None):
    """a [M,K] . b[N,K]^T -> [M,N]  (both K-contiguous: the MFMA layout).  b_scale: b holds e4m3 codes (see gemm)."""
    M, K = a.shape
    N = b.shape[0]
    c = out if out is not None else torch.empty(M, N, device=a.device, dtype=out_dtype or a.dtype)
    gemm(M, N, K, (a, 0), (a.stride(0), 1, 0, 0), (b, 0), (1, b.stride(0), 0, 0), (c, 0), (N, 1, 0, 0),
         bias=bias, residual=(residual, 0) if residual is not None else None,
         aux=(aux, 0) if aux is not None else None, alpha=alpha, epilogue=epilogue, b_scale=b_scale)
    return c


def mm_nn(a, b, aux=None, epilogue=EPI_NONE, alpha=1.0):
    """a [M,K] . b [K,N] -> [M,N] for row-major b (generic kernel unless the caller supplies b^T to mm_nt)."""
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, device=a.device, dtype=a.dtype)
    gemm(M, N, K, (a, 0), (a.stride(0), 1, 0, 0), (b, 0), (b.stride(0), 1, 0, 0), (c, 0), (N, 1, 0, 0),
         aux=(aux, 0) if aux is not None else None, epilogue=epilogue, alpha=alpha)
    return c


def transpose_pad(x, pad_to=64, dtype=None):
    """x [R,C] -> x^T zero-padded along R to a multiple of `pad_to`: [C, Rpad]."""
    R, Cc = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty(Cc, Rp, device=x.device, dtype=dtype or x.dtype)
    _lib.check(_lib.lib().focus_transpose_pad(_p(x), _dt(x), x.stride(0), 0, _p(out), _dt(out), Rp, 0, R, Cc, 1,
                                              _stream()), "transpose_pad")
    return out


def mm_tn(a, b):
    """a[M,N]^T . b[M,K] -> [N,K] fp32 (weight gradients).  bf16 with N, K multiples of 8: the TN MFMA kernels read both
    operands row-major (no transposed copies); other bf16 shapes are re-laid K(=M)-contiguous by focus_transpose_pad
    for the NT kernel; fp32: strided generic kernel."""
    M, N = a.shape
    K = b.shape[1]
    if a.dtype == torch.bfloat16 and USE_TN_GEMM and N % 8 == 0 and K % 8 == 0 and N >= 8 and K >= 8:
        # both operands stay row-major: the TN MFMA kernel gathers its fragments with transposed LDS reads; the long
        # reduction is split over workgroups, partial tiles go to slabs and are summed by a second kernel
        c = torch.empty(N, K, device=a.device, dtype=torch.float32)
        nb = _lib.lib().focus_gemm_tn_workspace_bytes(N, K, M)
        ws = torch.empty(nb // 4, device=a.device, dtype=torch.float32)
        gemm(N, K, M, (a, 0), (1, a.stride(0), 0, 0), (b, 0), (b.stride(0), 1, 0, 0), (c, 0), (K, 1, 0, 0),
             aux=(ws, 0))
    elif a.dtype == torch.bfloat16:
        at, bt = transpose_pad(a), transpose_pad(b)
        Mp = at.shape[1]
        c = torch.zeros(N, K, device=a.device, dtype=torch.float32)
        gemm(N, K, Mp, (at, 0), (Mp, 1, 0, 0), (bt, 0), (1, Mp, 0, 0), (c, 0), (K, 1, 0, 0), accumulate=True)
    else:
        c = torch.empty(N, K, device=a.device, dtype=torch.float32)
        gemm(N, K, M, (a, 0), (1, a.stride(0), 0, 0), (b, 0), (b.stride(0), 1, 0, 0), (c, 0), (K, 1, 0, 0))
    return c


import os as _os


def wgrad_async(dy, x, want_bias):
    """(dw, db, join) of a Linear.  (A side-stream variant that overlapped dW with dX was measured 4-5 % SLOWER on
    the bench step -- the two persistent kernels evict each other's operand panels from L2 -- and was removed.)"""
    dw, db = linear_wgrad(dy, x, want_bias)
    return dw, db, None


def linear_wgrad(dy, x, want_bias):
    """(dw [N,K] fp32, db [N] fp32 or None) for y = x @ w^T + b from dy [M,N], x [M,K]: one pass over dy
    (focus_linear_wgrad: the bias gradient rides on the weight-gradient GEMM's matrix pipe)."""
    M, N = dy.shape
    K = x.shape[1]
    ok = (dy.dtype == torch.bfloat16 and USE_TN_GEMM and N % 8 == 0 and K % 8 == 0 and N >= 8 and K >= 8
          and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0)
    if not ok:
        return mm_tn(dy, x), (colsum(dy) if want_bias else None)
    L = _lib.lib()
    dw = torch.empty(N, K, device=dy.device, dtype=torch.float32)
    db = torch.empty(N, device=dy.device, dtype=torch.float32) if want_bias else None
    nb = L.focus_linear_wgrad_workspace_bytes(N, K, M)
    ws = torch.empty(max(nb, 16) // 4, device=dy.device, dtype=torch.float32)
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(L.focus_linear_wgrad(_p(dy), _p(x), _p(dw), _p(db) if want_bias else None, _p(ws), nb, M, N, K,
                                    dy.stride(0), x.stride(0), _dt(dy), _stream()), "linear_wgrad")
    if GEMM_TIMING is not None:
        e1.record()
        GEMM_TIMING.append((2.0 * M * N * K, e0, e1, "tn", (N, K, M, 1, 0)))
    return dw, db


def colsum(x):
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().focus_colsum(_p(x), _p(out), M, N, x.stride(0), 0, _dt(x), _stream()), "colsum")
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    _lib.check(_lib.lib().focus_cast(_p(x), _dt(x), _p(out), _dt(out), x.numel(), _stream()), "cast")
    return out


# bf16 shadow copies of the fp32 master weights (and their transposes), refreshed when the parameter's
# version counter changes (optimizer steps are in-place).
_shadow_cache = {}
_shadow_gen = 0
_shadow_epoch = 0         # bumped whenever a shadow TENSOR is created or dropped (the fused optimizer caches their pointers)


def shadow_epoch():
    return _shadow_epoch


def invalidate_shadows():
    """Fused optimizers (torch._fused_adamw_) update parameters in place WITHOUT bumping Tensor._version, so the
    version counter alone cannot tell that a bf16 shadow is stale.  construct_optimizer() registers this as an
    optimizer post-step hook; call it yourself after any other out-of-band parameter update.
    The bf16 shadows that already exist (row-major and transposed) are rewritten right here by ONE multi-tensor
    launch (focus_shadow_refresh) instead of ~200 cast / transpose launches spread over the next step."""
    global _shadow_gen
    _shadow_gen += 1
    _refresh_shadows_batched()
    if _stacked:
        refresh_stacked()
    if _fp8_cache:
        refresh_fp8_shadows()


_shadow_tables = {}      # device -> (signature, items tensor, max_rows, max_cols, [(key, weakref)])


def _refresh_shadows_batched():
    import numpy as np
    by_dev = {}
    for key, (ref, stamp, out) in list(_shadow_cache.items()):
        w = ref()
        if w is None or key[1] != torch.bfloat16 or w.dtype != torch.float32 or w.dim() != 2 or not w.is_cuda:
            continue
        if w.shape[0] % 4 or w.shape[1] % 4 or not w.is_contiguous() or w.data_ptr() % 16:
            continue
        by_dev.setdefault(w.device, {}).setdefault(id(w), [w, None, None])[2 if key[2] else 1] = (key, out)
    for dev, ws in by_dev.items():
        sig = tuple((i, e[0].data_ptr(), e[1][1].data_ptr() if e[1] else 0, e[2][1].data_ptr() if e[2] else 0)
                    for i, e in sorted(ws.items()))
        tab = _shadow_tables.get(dev)
        if tab is None or tab[0] != sig:
            rec = np.zeros((len(sig), 4), dtype=np.int64)       # focus_shadow_item: 3 pointers + (rows, cols)
            mr = mc = 0
            for n, (i, sp, dp, tp) in enumerate(sig):
                w = ws[i][0]
                rec[n, 0], rec[n, 1], rec[n, 2] = sp, dp, tp
                rec[n, 3] = int(w.shape[0]) | (int(w.shape[1]) << 32)
                mr, mc = max(mr, w.shape[0]), max(mc, w.shape[1])
            tab = (sig, torch.from_numpy(rec).to(dev), mr, mc)
            _shadow_tables[dev] = tab
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().focus_shadow_refresh(_p(tab[1]), len(sig), tab[2], tab[3], _stream()), "shadow_refresh")
        for e in ws.values():
            w = e[0]
            for slot in (e[1], e[2]):
                if slot is not None:
                    key, out = slot
                    _shadow_cache[key] = (_shadow_cache[key][0], (w._version, w.data_ptr(), _shadow_gen), out)


def cached_shadow(w, dtype, transposed):
    """The existing shadow tensor of `w` (stale or not), or None: the fused optimizer step (optimizer.FusedAdamW)
    rewrites these in the same pass that updates the masters."""
    hit = _shadow_cache.get((id(w), dtype, transposed))
    if hit is not None and hit[0]() is w:
        return hit[2]
    return None


def shadows_written(entries):
    """entries: [(w, dst or None, dstT or None)] -- the bf16 shadows the optimizer step has just rewritten from the
    updated masters.  Starts a new shadow generation (every other cached copy is stale now) and stamps these as fresh."""
    global _shadow_gen
    _shadow_gen += 1
    for w, dst, dstT in entries:
        for out, tr in ((dst, False), (dstT, True)):
            if out is None:
                continue
            key = (id(w), torch.bfloat16, tr)
            hit = _shadow_cache.get(key)
            if hit is not None and hit[2] is out:
                _shadow_cache[key] = (hit[0], (w._version, w.data_ptr(), _shadow_gen), out)
    if _stacked:
        refresh_stacked()
    if _fp8_cache:
        refresh_fp8_shadows()


def drop_caches():
    """Forget every bf16 weight shadow (frees their memory when a model is dropped between workloads)."""
    global _shadow_epoch
    _shadow_epoch += 1
    _shadow_cache.clear()
    _shadow_tables.clear()
    _stacked.clear()
    _fp8_cache.clear()
    _fp8_tables.clear()


def shadow(w, dtype, transposed=False):
    w = getattr(w, "_focus_base", w)          # a wgrad_group alias stands for its parameter
    if dtype == torch.float32 and not transposed:
        return w.detach()
    key = (id(w), dtype, transposed)
    hit = _shadow_cache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr(), _shadow_gen):
        return hit[2]
    wd = w.detach()
    if transposed:
        N, K = wd.shape
        out = torch.empty(K, N, device=w.device, dtype=dtype)
        _lib.check(_lib.lib().focus_transpose_pad(_p(wd), _dt(wd), K, 0, _p(out), _dt(out), N, 0, N, K, 1, _stream()),
                   "shadow^T")
    else:
        out = cast(wd, dtype)
    global _shadow_epoch
    _shadow_epoch += 1
    _shadow_cache[key] = (weakref.ref(w, lambda _r, k=key: _shadow_cache.pop(k, None)),
                          (w._version, w.data_ptr(), _shadow_gen), out)
    return out


# --------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) working copies of the Linear weights: BASELINE configs[4] "fp8 MFMA weights", SURVEY 8(d) config 5
# "fp8 weights for the Linear GEMMs, bf16 activations".  One scale per tensor (amax / 448), the SAME codes in both
# orientations: the forward reads W_q [N,K], the dX GEMM its transpose [K,N]; the fp32 masters, the weight gradients
# (activations x activations) and everything else are untouched.  `with ops.fp8_weights(True):` around a forward makes
# its Linear / MLP nodes take the fp8 B operand (their backward follows what the forward used).
# --------------------------------------------------------------------------------------------------
FP8_WEIGHTS = False
_FP8_MIN_ROWS = 1024          # fewer activation rows: the bf16 small-row kernels (the fp8 instances are the large tiles)
_fp8_cache = {}               # id(w) -> [weakref, stamp, codes [N,K] uint8, codesT [K,N] uint8, scale [1] fp32]
_fp8_tables = {}              # device -> (signature, items tensor, max_rows, max_cols, amax scratch)


class fp8_weights:
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global FP8_WEIGHTS
        self.prev = FP8_WEIGHTS
        FP8_WEIGHTS = self.on
        return self

    def __exit__(self, *exc):
        global FP8_WEIGHTS
        FP8_WEIGHTS = self.prev
        return False


def fp8_ok(w, rows, dtype):
    w = getattr(w, "_focus_base", w)
    return (FP8_WEIGHTS and dtype == torch.bfloat16 and w.dim() == 2 and w.dtype == torch.float32 and w.is_cuda
            and rows >= _FP8_MIN_ROWS and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 and w.is_contiguous()
            and w.data_ptr() % 16 == 0)


def _fp8_run(dev, entries):
    import numpy as np
    sig = tuple((e[2].data_ptr(), e[3].data_ptr(), e[4].data_ptr(), w.data_ptr()) for w, e in entries)
    tab = _fp8_tables.get(dev)
    if tab is None or tab[0] != sig:
        rec = np.zeros((len(entries), 5), dtype=np.int64)          # focus_fp8_item: 4 pointers + (rows, cols)
        mr = mc = 0
        for n, (w, e) in enumerate(entries):
            rec[n, 0], rec[n, 1], rec[n, 2], rec[n, 3] = w.data_ptr(), e[2].data_ptr(), e[3].data_ptr(), e[4].data_ptr()
            rec[n, 4] = int(w.shape[0]) | (int(w.shape[1]) << 32)
            mr, mc = max(mr, w.shape[0]), max(mc, w.shape[1])
        tab = (sig, torch.from_numpy(rec).to(dev), mr, mc, torch.empty(len(entries), dtype=torch.int32, device=dev))
        if len(entries) > 1:
            _fp8_tables[dev] = tab
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().focus_fp8_refresh(_p(tab[1]), len(entries), tab[2], tab[3], _p(tab[4]), _stream()), "fp8_refresh")
    for w, e in entries:
        e[1] = (w._version, w.data_ptr(), _shadow_gen)


def refresh_fp8_shadows():
    """Requantise every cached fp8 weight copy from its fp32 master (one amax + one quantise launch per device): called
    after an optimizer step, like the bf16 shadows."""
    by_dev = {}
    for key, e in list(_fp8_cache.items()):
        w = e[0]()
        if w is None:
            _fp8_cache.pop(key, None)
            continue
        by_dev.setdefault(w.device, []).append((w, e))
    for dev, entries in by_dev.items():
        _fp8_run(dev, entries)


def shadow_fp8(w, transposed=False):
    """(e4m3 codes as a uint8 tensor -- [N,K], or [K,N] when transposed --, scale [1] fp32) of the 2-D weight w [N,K]."""
    w = getattr(w, "_focus_base", w)
    e = _fp8_cache.get(id(w))
    if e is None or e[0]() is not w:
        N, K = w.shape
        e = _fp8_cache[id(w)] = [weakref.ref(w, lambda _r, k=id(w): _fp8_cache.pop(k, None)), None,
                                 torch.empty(N, K, dtype=torch.uint8, device=w.device),
                                 torch.empty(K, N, dtype=torch.uint8, device=w.device),
                                 torch.empty(1, dtype=torch.float32, device=w.device)]
        _fp8_tables.pop(w.device, None)
    if e[1] != (w._version, w.data_ptr(), _shadow_gen):
        _fp8_run(w.device, [(w.detach(), e)])
        e[1] = (w._version, w.data_ptr(), _shadow_gen)
    return (e[3] if transposed else e[2]), e[4]


def _dx_from(dy, w, dtype, aux=None, epilogue=EPI_NONE, alpha=1.0, fp8=False):
    """alpha * dy [M,N] . w[N,K] -> [M,K].  fp8: the forward multiplied by the e4m3 copy of w, so does this."""
    if fp8:
        wq, sc = shadow_fp8(w, transposed=True)
        return mm_nt(dy, wq, aux=aux, epilogue=epilogue, alpha=alpha, b_scale=sc)
    if dtype == torch.bfloat16 and w.shape[0] % 64 == 0:
        return mm_nt(dy, shadow(w, dtype, transposed=True), aux=aux, epilogue=epilogue, alpha=alpha)
    return mm_nn(dy, shadow(w, dtype), aux=aux, epilogue=epilogue, alpha=alpha)


# --------------------------------------------------------------------------------------------------
# Deferred parameter gradients.  A recurrent module (STEVE's slot update: 24 frames x 3 iterations through the same
# GRU / MLP / LayerNorm parameters) applies every weight dozens of times per step; autograd would run one tiny
# weight-gradient GEMM, one bias column-sum and one accumulation add per application (~2000 launches of 5-13 us
# per STEVE step).  Inside `with deferred_wgrads():` the Linear / MLP / LayerNorm nodes recorded by the forward
# count their applications per parameter; in the backward every application only STASHES its (dY, X), and the node
# whose backward runs LAST (the count reaches zero) runs ONE weight-gradient GEMM over the stacked applications (the
# reduction over the applications becomes part of the GEMM's reduction dimension) and RETURNS it as its parameter
# gradient -- the others return None.  The sum reaches the parameter through autograd's own AccumulateGrad, so
# DistributedDataParallel's reducer hooks fire as for any other gradient.  Same sums, different association order.
# Valid when every application recorded by the forward takes part in the backward (true for the slot loop: the slots
# chain through all of them).  If some never run (loss on a subset of frames), the count never reaches zero: a
# callback at the end of the backward pass finds such leftovers and adds them to .grad directly (single process) or
# raises (multi-rank DDP, where a gradient must not bypass the reducer).
# --------------------------------------------------------------------------------------------------
_DEFER_ON = False
_DEFER_MAX_ROWS = 8192            # applications larger than this keep the immediate path (their GEMMs are efficient)
_defer_stashes = {}               # key -> _Stash, for the forward passes recorded since the last deferred_wgrads().__enter__
_leftover_queued = False


class _Stash:
    __slots__ = ("kind", "params", "alpha", "pending", "items")

    def __init__(self, kind, params, alpha):
        self.kind, self.params, self.alpha, self.pending, self.items = kind, params, alpha, 0, []


class deferred_wgrads:
    def __enter__(self):
        global _DEFER_ON, _defer_stashes, _leftover_queued
        self.prev = _DEFER_ON
        # a fresh table: stashes of an earlier forward whose backward never ran (or raised half way) are orphaned with
        # their graph instead of swallowing this step's gradients
        _defer_stashes = {}
        _leftover_queued = False
        _DEFER_ON = _os.environ.get("FOCUS_DEFER_WGRAD", "1") != "0"
        return self

    def __exit__(self, *exc):
        global _DEFER_ON
        _DEFER_ON = self.prev
        return False


def _defer_open(kind, params, rows, needed, alpha=1.0):
    """Forward side: the stash shared by all applications of `params` (a tuple of leaf parameters / None), or None when
    this application keeps the immediate path.  needed: ctx.needs_input_grad of the weight (False under no_grad and for
    frozen parameters: nothing is counted then, and nothing is written to a frozen parameter's .grad)."""
    if not _DEFER_ON or not needed or rows > _DEFER_MAX_ROWS:
        return None
    live = [p for p in params if p is not None]
    if not live or not all(p.is_leaf and p.requires_grad for p in live):
        return None
    key = (kind, alpha) + tuple(id(p) if p is not None else 0 for p in params)
    st = _defer_stashes.get(key)
    if st is None:
        st = _defer_stashes[key] = _Stash(kind, params, alpha)
    st.pending += 1
    return st


def _defer_finish(st):
    """The gradients of a complete stash: linear -> (dw, db or None); sum -> tuple of summed vectors; ln -> (dg, db)."""
    items, st.items = st.items, []
    if st.kind == "linear":
        dy = items[0][0] if len(items) == 1 else torch.cat([i[0] for i in items], 0)
        x = items[0][1] if len(items) == 1 else torch.cat([i[1] for i in items], 0)
        dw, db = linear_wgrad(dy, x, st.params[1] is not None)
        if st.alpha != 1.0:
            dw = dw * st.alpha
        return dw, db
    if st.kind == "ln":
        both = (items[0] if len(items) == 1 else torch.cat(items, 1)).sum(1)              # [2, D]
        return both[0], both[1]
    n = len(items[0])                                                                      # "sum": tuples of vectors
    return tuple(items[0][j] if len(items) == 1 else torch.stack([i[j] for i in items], 0).sum(0) for j in range(n))


def _defer_close(st, item):
    """Backward side: stash this application's contribution; the last one returns the finished gradients (a tuple, see
    _defer_finish), the others None."""
    global _leftover_queued
    st.items.append(item)
    st.pending -= 1
    if st.pending > 0:
        if not _leftover_queued:
            _leftover_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(_deferred_leftovers)
        return None
    return _defer_finish(st)


def _acc_grad(p, g):
    g = g.reshape(p.shape)
    if p.grad is None:
        p.grad = g.to(p.dtype) if g.dtype != p.dtype else g
    else:
        p.grad.add_(g)


def _deferred_leftovers():
    """End of a backward pass: stashes that hold contributions but whose count never reached zero (some application of
    the parameter took no part in this backward)."""
    global _leftover_queued
    _leftover_queued = False
    left = [st for st in _defer_stashes.values() if st.items]
    if not left:
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        raise RuntimeError("focus_amd.ops.deferred_wgrads: %d parameter(s) were applied more often in the forward than in this "
                           "backward; under DistributedDataParallel their gradient would bypass the reducer "
                           "(set FOCUS_DEFER_WGRAD=0 for this loss)" % len(left))
    with torch.no_grad():
        for st in left:
            st.pending = 0
            for p, g in zip([q for q in st.params if q is not None], [g for g in _defer_finish(st) if g is not None]):
                _acc_grad(p, g)


# --------------------------------------------------------------------------------------------------
# Grouped weight gradients.  The Linear layers of one Motionformer / ORViT block (qkv, proj_q, proj, fc1, fc2, ...) each
# give 18-72 output tiles of dW with a 12552-row reduction: alone none of them fills the 256 CUs without splitting the
# reduction into slabs that a second launch sums.  Together they are ~234 tiles = one round of the machine with no split.
# `with ops.wgrad_group(params):` around a block's forward routes the parameters through ONE autograd node (_ParamGroupFn)
# that sits upstream of every Linear / MLP node using them: those nodes stash (dY, X) instead of computing dW, and the
# group node -- which autograd runs after all of them -- forms every dW and db of the block in one launch
# (focus_linear_wgrad_group, csrc/gemm_tn_group.hip) and returns them as ITS gradients: they reach the parameters
# through AccumulateGrad like any other gradient (DistributedDataParallel's hooks fire unchanged).
# --------------------------------------------------------------------------------------------------
_GROUP_ON = _os.environ.get("FOCUS_WGRAD_GROUP", "1") != "0"
_GROUP_MIN_ROWS = 4096
_GROUP_MIN_UNITS = 128
_group_stack = []
_group_tables = {}      # (device, (N, K) of the problems of a launch) -> device unit list (focus_linear_wgrad_group_plan)


class _WgradGroup:
    def __init__(self, params):
        self.params = params
        self.index = {id(p): i for i, p in enumerate(params)}
        self.alias = {}                 # id(param) -> aliased tensor produced by the group node
        self.stash = []                 # (index of w, index of b or -1, dy2, x2)


class _ParamGroupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grp, *params):
        ctx.set_materialize_grads(False)
        ctx.grp = grp
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *gs):
        grp = ctx.grp
        out = list(gs)
        stash, grp.stash = grp.stash, []
        if stash:
            for i, g in _group_flush(grp, stash).items():
                out[i] = g if out[i] is None else out[i] + g
        return (None, *out)


def _group_flush(grp, stash):
    """All dW / db of the stashed (dY, X) pairs: one grouped launch per 8 problems when they fill the machine, the
    per-Linear path otherwise.  -> {param index: gradient}."""
    L = _lib.lib()
    res = {}
    # several applications of one weight inside the group (not the case in the blocks): stack them
    by_w = {}
    for wi, bi, dy2, x2 in stash:
        by_w.setdefault((wi, bi), []).append((dy2, x2))
    probs = []
    for (wi, bi), lst in by_w.items():
        dy2 = lst[0][0] if len(lst) == 1 else torch.cat([a for a, _ in lst], 0)
        x2 = lst[0][1] if len(lst) == 1 else torch.cat([b for _, b in lst], 0)
        probs.append((wi, bi, dy2, x2))
    probs.sort(key=lambda t: -t[2].shape[0])
    # one unit streams the WHOLE reduction of its tile: a problem whose reduction is much longer than the others' (ORViT's
    # patch_to_d[2] on the 4x more RoI-cell rows) would hold a few CUs long after the rest has finished -- it keeps the
    # per-Linear path, which splits its reduction over the machine
    med = sorted(t[2].shape[0] for t in probs)[len(probs) // 2]
    for t in [t for t in probs if t[2].shape[0] > 1.5 * med]:
        probs.remove(t)
        dw, db = linear_wgrad(t[2], t[3], t[1] >= 0)
        res[t[0]] = dw
        if t[1] >= 0:
            res[t[1]] = db
    Item = _lib.WgradItem
    arr = (Item * len(probs))()
    for n, (wi, bi, dy2, x2) in enumerate(probs):
        arr[n].M, arr[n].N, arr[n].K = dy2.shape[0], dy2.shape[1], x2.shape[1]
    units = L.focus_linear_wgrad_group_units(arr, len(probs))
    if units < _GROUP_MIN_UNITS:
        for wi, bi, dy2, x2 in probs:
            dw, db = linear_wgrad(dy2, x2, bi >= 0)
            res[wi] = dw
            if bi >= 0:
                res[bi] = db
        return res
    dev = probs[0][2].device
    nw = sum(t[2].shape[1] * t[3].shape[1] for t in probs)
    nb = sum(t[2].shape[1] for t in probs if t[1] >= 0)
    flat = torch.empty(nw + nb, device=dev, dtype=torch.float32)
    if nb:
        flat[nw:].zero_()                                    # the bias sums are accumulated with atomics
    ow, ob = 0, nw
    for n, (wi, bi, dy2, x2) in enumerate(probs):
        M, N = dy2.shape
        K = x2.shape[1]
        dw = flat[ow:ow + N * K].view(N, K)
        ow += N * K
        res[wi] = dw
        arr[n].dy, arr[n].x, arr[n].dw = dy2.data_ptr(), x2.data_ptr(), dw.data_ptr()
        arr[n].ld_dy, arr[n].ld_x = dy2.stride(0), x2.stride(0)
        arr[n].db = None
        if bi >= 0:
            db = flat[ob:ob + N]
            ob += N
            res[bi] = db
            arr[n].db = db.data_ptr()
    if GEMM_TIMING is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # at most 8 problems per launch: more are dealt to ceil(n/8) launches of about equal size (largest first)
    nch = (len(probs) + 7) // 8
    chunks = [[] for _ in range(nch)]
    load = [0] * nch
    per = [((arr[j].N + 255) // 256) * ((arr[j].K + 127) // 128) for j in range(len(probs))]
    for j in sorted(range(len(probs)), key=lambda j: -per[j]):
        c = min((c for c in range(nch) if len(chunks[c]) < 8), key=lambda c: load[c])
        chunks[c].append(j)
        load[c] += per[j]
    for ch in chunks:
        ch.sort()                                            # keeps the longest reductions first inside a launch
        sub = (Item * len(ch))(*[arr[j] for j in ch])
        sig = (dev,) + tuple((arr[j].N, arr[j].K) for j in ch)
        table = _group_tables.get(sig)
        if table is None:                                    # the unit list depends on the (N, K) sequence only
            nu = L.focus_linear_wgrad_group_units(sub, len(ch))
            host = torch.empty(nu * 8, dtype=torch.uint8).pin_memory()
            _lib.check(L.focus_linear_wgrad_group_plan(sub, len(ch), ctypes.c_void_p(host.data_ptr()), nu * 8), "wgrad_group_plan")
            table = _group_tables[sig] = host.to(dev)        # (blocking copy, once per signature)
        _lib.check(L.focus_linear_wgrad_group(sub, len(ch), _p(table), _stream()), "linear_wgrad_group")
    if GEMM_TIMING is not None:
        e1.record()
        fl = sum(2.0 * t[2].shape[0] * t[2].shape[1] * t[3].shape[1] for t in probs)
        GEMM_TIMING.append((fl, e0, e1, "tn", (len(probs), units, probs[0][2].shape[0], 1, 0)))
    return res


class wgrad_group:
    """with ops.wgrad_group(parameters of one block): ... the block's forward ...   (see the section comment above)"""

    def __init__(self, params):
        self.params = [p for p in params if p is not None]

    def __enter__(self):
        self.grp = None
        ps = self.params
        if (_GROUP_ON and torch.is_grad_enabled() and ps and all(p.requires_grad and p.is_cuda for p in ps)):
            self.grp = _WgradGroup(ps)
            for p, a in zip(ps, _ParamGroupFn.apply(self.grp, *ps)):
                a._focus_base = p
                self.grp.alias[id(p)] = a
            _group_stack.append(self.grp)
        return self

    def __exit__(self, *exc):
        if self.grp is not None:
            _group_stack.pop()
            # the aliases are referenced by the autograd graph from here on; keeping them in the group (which the group
            # node's ctx holds) would close a reference cycle through C++ autograd nodes that the collector cannot break
            self.grp.alias = {}
            self.grp = None
        return False


def _grouped(t):
    """The active group's alias of a parameter (or the tensor itself)."""
    if t is not None and _group_stack:
        a = _group_stack[-1].alias.get(id(t))
        if a is not None:
            return a
    return t


def _group_of(w, b, rows, dtype):
    """(group, index of w, index of b) when this Linear's weight gradient is left to the active group."""
    if not _group_stack or dtype != torch.bfloat16 or rows < _GROUP_MIN_ROWS or not USE_TN_GEMM:
        return None
    grp = _group_stack[-1]
    bw = getattr(w, "_focus_base", None)
    if bw is None or id(bw) not in grp.index or w.dim() != 2 or w.shape[0] % 8 or w.shape[1] % 8:
        return None
    bi = -1
    if b is not None:
        bb = getattr(b, "_focus_base", None)
        if bb is None or id(bb) not in grp.index:
            return None
        bi = grp.index[id(bb)]
    return grp, grp.index[id(bw)], bi


# --------------------------------------------------------------------------------------------------
# Linear / MLP
# --------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, residual, alpha):
        _need_gpu(x, w)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = residual.reshape(-1, w.shape[0]).contiguous() if residual is not None else None
        ctx.fp8 = fp8_ok(w, x2.shape[0], x.dtype)
        if ctx.fp8:
            wq, sc = shadow_fp8(w)
            y = mm_nt(x2, wq, bias=b, residual=r2, alpha=alpha, b_scale=sc)
        else:
            y = mm_nt(x2, shadow(w, x.dtype), bias=b, residual=r2, alpha=alpha)
        ctx.save_for_backward(x2, w)
        ctx.has_b, ctx.has_r, ctx.shp, ctx.alpha = b is not None, residual is not None, shp, alpha
        ctx.defer = _defer_open("linear", (w, b), x2.shape[0], ctx.needs_input_grad[1], alpha)
        ctx.group = None
        ctx.wbase = getattr(w, "_focus_base", w)               # shadows are keyed by the parameter, not by a group alias
        if ctx.defer is None and alpha == 1.0 and ctx.needs_input_grad[1] and x2.stride(0) % 8 == 0:
            ctx.group = _group_of(w, b, x2.shape[0], x.dtype)
        return y.reshape(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        join = None
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        if ctx.defer is not None:
            done = _defer_close(ctx.defer, (dy2, x2))
            if done is not None:
                dw, db = done
        elif ctx.group is not None:
            grp, wi, bi = ctx.group
            grp.stash.append((wi, bi if want_b else -1, dy2, x2))      # the group node forms dW / db for the whole block
        elif ctx.needs_input_grad[1]:
            dw, db, join = wgrad_async(dy2, x2, want_b)
            if ctx.alpha != 1.0:
                dw = dw * ctx.alpha
        elif want_b:
            db = colsum(dy2)
        if ctx.needs_input_grad[0]:
            dx = _dx_from(dy2, ctx.wbase, dy2.dtype, alpha=ctx.alpha, fp8=ctx.fp8).reshape(ctx.shp)
        if join is not None:
            join()
        return dx, dw, db, (dy if ctx.has_r else None), None


def linear(x, w, b=None, residual=None, alpha=1.0):
    """nn.Linear forward (+ fused residual add): y = alpha * x.w^T + b (+ residual)."""
    return _LinearFn.apply(x, _grouped(w), _grouped(b), residual, float(alpha))


class _LinearKVFn(torch.autograd.Function):
    """Two bias-free Linears of the same input (STEVE's k / v projections of a frame, steve.py:62-63).  Forward is the
    two GEMMs; the backward forms d(input) = alpha dk.Wk + dv.Wv with the second product's epilogue adding the first
    (autograd would write both [rows, Din] products and add them in a third pass: 150 MB of traffic per frame)."""

    @staticmethod
    def forward(ctx, x, wk, wv, alpha_k):
        _need_gpu(x, wk, wv)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        k = mm_nt(x2, shadow(wk, x.dtype), alpha=alpha_k)
        v = mm_nt(x2, shadow(wv, x.dtype))
        ctx.save_for_backward(x2, wk, wv)
        ctx.shp, ctx.alpha = shp, alpha_k
        return k.reshape(*shp[:-1], wk.shape[0]), v.reshape(*shp[:-1], wv.shape[0])

    @staticmethod
    def backward(ctx, dk, dv):
        x2, wk, wv = ctx.saved_tensors
        Dk, Dv = wk.shape[0], wv.shape[0]
        dk2, dv2 = dk.reshape(-1, Dk), dv.reshape(-1, Dv)
        dt = dk2.dtype
        need_x, need_wk, need_wv = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if (Dk == Dv and dt == torch.bfloat16 and Dk % 64 == 0 and dk2.stride() == (2 * Dk, 1) == dv2.stride()
                and dv2.data_ptr() == dk2.data_ptr() + Dk * dk2.element_size()):
            # the two gradients are the halves of one [rows, 2D] matrix (ops.slot_attn_step's deferred k/v gradient):
            # d(input) = [dk|dv] . [alpha Wk; Wv] and d[Wk; Wv] = [dk|dv]^T . x, each ONE product that reads [dk|dv] once
            dkv = torch.as_strided(dk2, (dk2.shape[0], 2 * Dk), (2 * Dk, 1))
            dx = dwk = dwv = None
            if need_x:
                dx = mm_nt(dkv, _stacked_wT(wk, wv, ctx.alpha, dt)).reshape(ctx.shp)
            if need_wk or need_wv:
                dw = linear_wgrad(dkv, x2, False)[0]
                dwk = dw[:Dk] * ctx.alpha if ctx.alpha != 1.0 else dw[:Dk]
                dwv = dw[Dk:]
            return dx, dwk if need_wk else None, dwv if need_wv else None, None
        dk2 = dk2 if dk2.is_contiguous() else dk2.contiguous()
        dv2 = dv2 if dv2.is_contiguous() else dv2.contiguous()
        dwk = dwv = dx = None
        if need_wk:
            dwk = linear_wgrad(dk2, x2, False)[0]
            if ctx.alpha != 1.0:
                dwk = dwk * ctx.alpha
        if need_wv:
            dwv = linear_wgrad(dv2, x2, False)[0]
        if need_x:
            if dt == torch.bfloat16 and Dk % 64 == 0 and Dv % 64 == 0:
                dx = mm_nt(dk2, shadow(wk, dt, transposed=True), alpha=ctx.alpha)
                mm_nt(dv2, shadow(wv, dt, transposed=True), residual=dx, out=dx)   # (each element: read, add, write by one lane)
            else:
                dx = _dx_from(dk2, wk, dt, alpha=ctx.alpha) + _dx_from(dv2, wv, dt)
            dx = dx.reshape(ctx.shp)
        return dx, dwk, dwv, None


_stacked = {}       # key -> [stamp, tensor, weak refs of the weights, build(out=None or the cached tensor) -> tensor]


def _stacked_get(key, ws, extra, build):
    """A weight-derived GEMM operand built from several bf16 shadows (concatenations / stacks), cached like the shadows
    themselves: valid for (version, storage) of every weight and the shadow generation.  A stale entry of unchanged shape
    is REBUILT IN PLACE (same storage): a captured HIP graph that reads the operand keeps reading the right address, and
    refresh_stacked() -- called from the optimizer's post-step hook -- does that rebuild eagerly for every live entry."""
    stamp = tuple(x for w in ws for x in (w._version, w.data_ptr())) + (extra, _shadow_gen)
    e = _stacked.get(key)
    if e is None or e[0] != stamp:
        if e is not None and all(r() is w for r, w in zip(e[2], ws)):
            t = build(e[1])
        else:
            t = build(None)
        e = _stacked[key] = [stamp, t, [weakref.ref(w) for w in ws], build]
    return e[1]


def refresh_stacked():
    """Rebuild every cached stacked operand in place from the (already refreshed) shadows."""
    for key, e in list(_stacked.items()):
        ws = [r() for r in e[2]]
        if any(w is None for w in ws):
            _stacked.pop(key, None)
            continue
        e[1] = e[3](e[1])
        e[0] = tuple(x for w in ws for x in (w._version, w.data_ptr())) + (e[0][-2], _shadow_gen)


def _stacked_wT(wk, wv, alpha, dtype):
    """[alpha Wk; Wv]^T as an NT-GEMM B operand [Din, 2D] in `dtype`."""
    def build(out):
        parts = [shadow(wk, dtype, transposed=True) * alpha, shadow(wv, dtype, transposed=True)]
        return torch.cat(parts, dim=1, out=out) if out is not None else torch.cat(parts, dim=1).contiguous()
    return _stacked_get((id(wk), id(wv), dtype), (wk, wv), alpha, build)


def linear_kv(x, wk, wv, alpha_k=1.0):
    """(alpha_k * x.wk^T, x.wv^T): two bias-free Linears of one input with a fused d(input)."""
    return _LinearKVFn.apply(x, wk, wv, float(alpha_k))


class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, act):
        _need_gpu(x, w1, w2)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        H = w1.shape[0]
        z = torch.empty(M, H, device=x.device, dtype=x.dtype) if act == EPI_GELU else None
        ctx.fp8 = fp8_ok(w1, M, x.dtype) and fp8_ok(w2, M, x.dtype)
        r2 = residual.reshape(-1, w2.shape[0]).contiguous() if residual is not None else None
        if ctx.fp8:
            (q1, s1), (q2, s2) = shadow_fp8(w1), shadow_fp8(w2)
            a = mm_nt(x2, q1, bias=b1, aux=z, epilogue=act, b_scale=s1)
            y = mm_nt(a, q2, bias=b2, residual=r2, b_scale=s2)
        else:
            a = mm_nt(x2, shadow(w1, x.dtype), bias=b1, aux=z, epilogue=act)
            y = mm_nt(a, shadow(w2, x.dtype), bias=b2, residual=r2)
        ctx.save_for_backward(x2, w1, w2, a, z)
        ctx.act, ctx.shp = act, shp
        ctx.has = (b1 is not None, b2 is not None, residual is not None)
        d1 = _defer_open("linear", (w1, b1), M, ctx.needs_input_grad[1])
        d2 = _defer_open("linear", (w2, b2), M, ctx.needs_input_grad[3]) if d1 is not None else None
        if d1 is not None and d2 is None:
            d1.pending -= 1
            d1 = None
        ctx.defer = (d1, d2) if d1 is not None else None
        ctx.group = None
        ctx.wbase = (getattr(w1, "_focus_base", w1), getattr(w2, "_focus_base", w2))
        if ctx.defer is None and ctx.needs_input_grad[1] and ctx.needs_input_grad[3] and x2.stride(0) % 8 == 0:
            g1, g2 = _group_of(w1, b1, M, x.dtype), _group_of(w2, b2, M, x.dtype)
            if g1 is not None and g2 is not None:
                ctx.group = (g1, g2)
        return y.reshape(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, a, z = ctx.saved_tensors
        dy2 = dy.reshape(-1, w2.shape[0])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        join2 = join1 = None
        dfr = ctx.defer
        grp = ctx.group
        if dfr is not None:
            dw2 = db2 = None
            done = _defer_close(dfr[1], (dy2, a))
            if done is not None:
                dw2, db2 = done
        elif grp is not None:
            dw2 = db2 = None
            grp[1][0].stash.append((grp[1][1], grp[1][2] if ctx.has[1] else -1, dy2, a))
        elif ctx.needs_input_grad[3]:
            dw2, db2, join2 = wgrad_async(dy2, a, ctx.has[1])
        else:
            dw2, db2 = None, (colsum(dy2) if ctx.has[1] else None)
        # dz = (dy . w2) * act'(.) fused in the GEMM epilogue
        dz = _dx_from(dy2, ctx.wbase[1], dy2.dtype, aux=(z if ctx.act == EPI_GELU else a), epilogue=_DEPI[ctx.act], fp8=ctx.fp8)
        if dfr is not None:
            dw1 = db1 = None
            done = _defer_close(dfr[0], (dz, x2))
            if done is not None:
                dw1, db1 = done
        elif grp is not None:
            dw1 = db1 = None
            grp[0][0].stash.append((grp[0][1], grp[0][2] if ctx.has[0] else -1, dz, x2))
        elif ctx.needs_input_grad[1]:
            dw1, db1, join1 = wgrad_async(dz, x2, ctx.has[0])
        else:
            dw1, db1 = None, (colsum(dz) if ctx.has[0] else None)
        dx = _dx_from(dz, ctx.wbase[0], dz.dtype, fp8=ctx.fp8).reshape(ctx.shp) if ctx.needs_input_grad[0] else None
        for j in (join2, join1):
            if j is not None:
                j()
        return dx, dw1, db1, dw2, db2, (dy if ctx.has[2] else None), None


def mlp(x, w1, b1, w2, b2, residual=None, act=EPI_GELU):
    """Linear -> act -> Linear (+ residual): common.py:26-34 (GELU), steve.py:46-49 / transformer.py:62-66 (ReLU)."""
    return _MlpFn.apply(x, _grouped(w1), _grouped(b1), _grouped(w2), _grouped(b2), residual, act)


class _ScaleAddFn(torch.autograd.Function):
    """out = x + scale[b] * y  (per-sample scale: stochastic depth on a residual branch), one pass each way."""

    @staticmethod
    def forward(ctx, x, y, scale, keep):
        _need_gpu(x, y, scale)
        x, y = x.contiguous(), y.contiguous()
        B = x.shape[0]
        per = x.numel() // B
        out = torch.empty_like(x)
        _lib.check(_lib.lib().focus_scale_add(_p(x), _p(y), _p(scale), keep, _p(out), B, per, _dt(x), _stream()),
                   "scale_add")
        ctx.save_for_backward(scale)
        ctx.keep = keep
        return out

    @staticmethod
    def backward(ctx, dout):
        (scale,) = ctx.saved_tensors
        dout = dout.contiguous()
        B = dout.shape[0]
        dy = torch.empty_like(dout)
        _lib.check(_lib.lib().focus_scale_add(None, _p(dout), _p(scale), ctx.keep, _p(dy), B, dout.numel() // B,
                                              _dt(dout), _stream()), "scale_add_bwd")
        return dout, dy, None, None


def residual_drop_path(x, y, drop_prob, training):
    """x + drop_path(y) (common.py:46-60): the per-sample keep mask is drawn exactly as the reference draws it."""
    if drop_prob == 0.0 or not training:
        return x + y
    keep = 1.0 - drop_prob
    # mask = floor(keep + u) and mask / keep are formed inside the kernel from the draws u
    return _ScaleAddFn.apply(x, y, torch.rand(x.shape[0], dtype=torch.float32, device=x.device), keep)


# --------------------------------------------------------------------------------------------------
# LayerNorm
# --------------------------------------------------------------------------------------------------
def _ln_forward(ctx, x, gamma, beta, eps):
    _need_gpu(x, gamma)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    rows = x2.shape[0]
    y = torch.empty_like(x2)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().focus_layernorm_fwd(_p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D,
                                              eps, _dt(x2), _stream()), "layernorm_fwd")
    ctx.save_for_backward(x2, gamma, mean, rstd)
    ctx.shp = x.shape
    ctx.defer = _defer_open("ln", (gamma, beta), rows, ctx.needs_input_grad[1])
    return y.reshape(x.shape)


def _ln_backward(ctx, dy, dres):
    x2, gamma, mean, rstd = ctx.saved_tensors
    D = x2.shape[1]
    rows = x2.shape[0]
    dy2 = dy.reshape(-1, D)
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    r2 = None
    if dres is not None:
        r2 = dres.reshape(-1, D)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    L = _lib.lib()
    nblk = L.focus_layernorm_bwd_blocks(rows)
    partial = torch.empty(2, nblk, D, device=x2.device, dtype=torch.float32)
    dx = torch.empty_like(x2)
    if ctx.defer is not None:
        # recurrent module: this application only leaves its block partials; every application's partials are summed in
        # one pass at the end of the backward (no per-application finish launch)
        _lib.check(L.focus_layernorm_bwd(_p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(r2) if r2 is not None else None,
                                         _p(dx), None, None, _p(partial), rows, D, _dt(x2), _stream()), "layernorm_bwd")
        done = _defer_close(ctx.defer, partial)
        return (dx.reshape(ctx.shp),) + (done if done is not None else (None, None))
    dg = torch.empty(D, device=x2.device, dtype=torch.float32)
    db = torch.empty(D, device=x2.device, dtype=torch.float32)
    _lib.check(L.focus_layernorm_bwd(_p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(r2) if r2 is not None else None,
                                     _p(dx), _p(dg), _p(db), _p(partial), rows, D, _dt(x2), _stream()), "layernorm_bwd")
    return dx.reshape(ctx.shp), dg, db


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        return _ln_forward(ctx, x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        return (*_ln_backward(ctx, dy, None), None)


class _LayerNormForkFn(torch.autograd.Function):
    """(x, LayerNorm(x)) for a pre-norm residual block: the gradient that comes back on the residual path is added
    to the LayerNorm's input gradient inside focus_layernorm_bwd instead of by a separate autograd accumulation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        return x.view_as(x), _ln_forward(ctx, x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dres, dy):
        if dy is None:
            return dres, None, None, None
        return (*_ln_backward(ctx, dy, dres), None)


class _UnbindFramesFn(torch.autograd.Function):
    """x [B,T,...] -> T contiguous frames.  Slicing `x[:, t]` per frame makes autograd build, for every frame, a
    zero-filled whole-video gradient and add it to the running sum (24 x (1.2 GB fill + 3.6 GB add) per STEVE step
    at the BASELINE shape: 18.6 ms of 77); here the T frame gradients are stacked once."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape, ctx.dtype, ctx.device = x.shape, x.dtype, x.device
        return tuple(x[:, t].contiguous() for t in range(x.shape[1]))

    @staticmethod
    def backward(ctx, *grads):
        B, T = ctx.shape[0], ctx.shape[1]
        out = torch.empty(ctx.shape, dtype=ctx.dtype, device=ctx.device)
        for t, g in enumerate(grads):
            if g is None:
                out[:, t].zero_()
            else:
                out[:, t].copy_(g)
        return out


def unbind_frames(x):
    """Frames of a [B,T,...] video tensor as T contiguous tensors (see _UnbindFramesFn)."""
    return _UnbindFramesFn.apply(x)


def _guard_shared(shared, what):
    """The shared-gradient nodes (FrameGrad, SlotKVGrad) hand autograd ONE gradient from the node whose backward runs last.
    That needs every node that was counted in forward to run in this backward; a loss on a subset of the frames /
    iterations (or torch.autograd.grad on an early output) leaves some unreached, and the shared gradient would silently
    never leave.  The first node of a backward pass therefore queues a check for the end of that pass: nodes still pending
    then raise instead of dropping d(inputs)."""
    if getattr(shared, "_guarded", False):
        return
    shared._guarded = True

    def check():
        shared._guarded = False
        if shared.pending != 0:
            left = shared.pending
            shared.pending = 0
            for n in ("buf", "dk", "dv"):
                if hasattr(shared, n):
                    setattr(shared, n, None)
            if hasattr(shared, "items"):
                shared.items = []
            raise RuntimeError("%s: %d of the nodes sharing this gradient were not reached by backward (a loss on a subset "
                               "of the frames / iterations?); their shared gradient would have been dropped -- give such "
                               "uses their own %s, or none" % (what, left, type(shared).__name__))
    torch.autograd.Variable._execution_engine.queue_callback(check)


class FrameGrad:
    """Whole-video gradient buffer shared by the per-frame LayerNorm nodes of one video tensor (see layer_norm_frame)."""

    def __init__(self):
        self.buf = None
        self.pending = 0
        self.frames = 0


class _FrameLayerNormFn(torch.autograd.Function):
    """LayerNorm of frame t of a [B,T,N,D] video tensor, read in place (steve.py:60 normalises the whole video, the
    slot loop consumes one frame at a time).  Forward: no x[:, t].contiguous() copy.  Backward: every frame's node
    writes its d(x) rows straight into ONE whole-video buffer (focus_layernorm_bwd_blocks_strided); the node that runs
    last hands the buffer to autograd, the others return None -- no per-frame copy, no zero-filled whole-video
    gradients summed by autograd (all T frames must be normalised through the same FrameGrad, which the slot loop does;
    frames that never ran are zero-filled)."""

    @staticmethod
    def forward(ctx, video, t, gamma, beta, eps, shared):
        _need_gpu(video, gamma)
        assert video.dim() == 4 and video.is_contiguous()
        B, T, N, D = video.shape
        rows = B * N
        y = torch.empty(B, N, D, device=video.device, dtype=video.dtype)
        mean = torch.empty(rows, device=video.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=video.device, dtype=torch.float32)
        _lib.check(_lib.lib().focus_layernorm_fwd_blocks(_p(video, t * N * D), N, T * N * D, _p(gamma), _p(beta), _p(y),
                                                         _p(mean), _p(rstd), rows, D, eps, _dt(video), _stream()),
                   "layernorm_fwd_blocks")
        ctx.save_for_backward(video, gamma, mean, rstd)
        ctx.t, ctx.shared = t, shared
        ctx.defer = _defer_open("sum", (gamma, beta), 0, ctx.needs_input_grad[2])
        shared.pending += 1
        shared.frames |= 1 << t
        return y

    @staticmethod
    def backward(ctx, dy):
        video, gamma, mean, rstd = ctx.saved_tensors
        B, T, N, D = video.shape
        rows, t, sh = B * N, ctx.t, ctx.shared
        _guard_shared(sh, "layer_norm_frame")
        dy = dy.contiguous()
        if sh.buf is None:
            sh.buf = torch.empty_like(video)
            for tt in range(T):
                if not (sh.frames >> tt) & 1:
                    sh.buf[:, tt].zero_()
        L = _lib.lib()
        nblk = L.focus_layernorm_bwd_blocks(rows)
        partial = torch.empty(2, nblk, D, device=video.device, dtype=torch.float32)
        dg = torch.empty(D, device=video.device, dtype=torch.float32)
        db = torch.empty(D, device=video.device, dtype=torch.float32)
        _lib.check(L.focus_layernorm_bwd_blocks_strided(_p(dy), _p(video, t * N * D), N, T * N * D, _p(gamma), _p(mean),
                                                        _p(rstd), _p(sh.buf, t * N * D), _p(dg), _p(db), _p(partial), rows, D,
                                                        _dt(video), _stream()), "layernorm_bwd_blocks")
        sh.pending -= 1
        out = None
        if sh.pending == 0:
            out, sh.buf, sh.frames = sh.buf, None, 0
        if ctx.defer is not None:
            done = _defer_close(ctx.defer, (dg, db))
            dg, db = done if done is not None else (None, None)
        return out, None, dg, db, None, None


def layer_norm_frame(video, t, gamma, beta, eps, shared):
    """LayerNorm(video[:, t]) -> [B,N,D]; `shared`: one FrameGrad for all frames of `video`."""
    return _FrameLayerNormFn.apply(video, t, gamma, beta, eps, shared)


def layer_norm(x, gamma, beta, eps):
    return _LayerNormFn.apply(x, gamma, beta, eps)


def layer_norm_fork(x, gamma, beta, eps):
    """-> (x_res, LayerNorm(x)): use x_res for the block's residual add (see _LayerNormForkFn)."""
    return _LayerNormForkFn.apply(x, gamma, beta, eps)


# --------------------------------------------------------------------------------------------------
# Trajectory attention
# --------------------------------------------------------------------------------------------------
class _TrajSpaceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, F_, P, heads):
        _need_gpu(qkv)
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        C = C3 // 3
        d = C // heads
        S = F_ * P
        assert N == S + 1
        dev, dt = qkv.device, qkv.dtype
        xt = torch.empty(B, S, F_, C, device=dev, dtype=dt)
        xdiag = torch.empty(B, S, C, device=dev, dtype=dt)
        cls_out = torch.empty(B, 1, C, device=dev, dtype=dt)
        lse = torch.empty(B, heads, S, F_, device=dev, dtype=torch.float32)
        cls_lse = torch.empty(B, heads, device=dev, dtype=torch.float32)
        L = _lib.lib()
        nb = L.focus_traj_space_workspace_bytes(B, F_, P, heads, d, _dt(qkv), 0)
        ws = torch.empty(max(nb, 16), device=dev, dtype=torch.uint8)
        _lib.check(L.focus_traj_space_fwd(_p(qkv), _p(xt), _p(xdiag), _p(cls_out), _p(lse), _p(cls_lse), _p(ws), nb,
                                          B, F_, P, heads, d, _dt(qkv), _stream()), "traj_space_fwd")
        ctx.save_for_backward(qkv, xt, cls_out, lse, cls_lse)
        ctx.dims = (B, F_, P, heads, d)
        return xt, xdiag, cls_out

    @staticmethod
    def backward(ctx, dxt, dxdiag, dcls):
        qkv, xt, cls_out, lse, cls_lse = ctx.saved_tensors
        B, F_, P, heads, d = ctx.dims
        dxt, dxdiag, dcls = dxt.contiguous(), dxdiag.contiguous(), dcls.contiguous()
        dqkv = torch.empty_like(qkv)
        L = _lib.lib()
        nb = L.focus_traj_space_workspace_bytes(B, F_, P, heads, d, _dt(qkv), 1)
        ws = torch.empty(max(nb, 16), device=qkv.device, dtype=torch.uint8)
        _lib.check(L.focus_traj_space_bwd(_p(qkv), _p(xt), _p(cls_out), _p(lse), _p(cls_lse), _p(dxt), _p(dxdiag),
                                          _p(dcls), _p(dqkv), _p(ws), nb, B, F_, P, heads, d, _dt(qkv), _stream()),
                   "traj_space_bwd")
        return dqkv, None, None, None


def traj_space(qkv, F_, P, heads):
    """(x~ [B,S,F,C], x_diag [B,S,C], cls_out [B,1,C]) from the fused qkv projection [B,1+F*P,3C]."""
    return _TrajSpaceFn.apply(qkv, F_, P, heads)


class _TrajTimeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q2, k2, xt, heads):
        _need_gpu(q2, k2, xt)
        q2, k2, xt = q2.contiguous(), k2.contiguous(), xt.contiguous()
        B, S, F_, C = xt.shape
        d = C // heads
        out = torch.empty(B, S, C, device=xt.device, dtype=xt.dtype)
        attn2 = torch.empty(B, heads, S, F_, device=xt.device, dtype=torch.float32)
        _lib.check(_lib.lib().focus_traj_time_fwd(_p(q2), _p(k2), _p(xt), _p(out), S * C, _p(attn2), B, S, F_, heads, d,
                                                  _dt(xt), _stream()), "traj_time_fwd")
        ctx.save_for_backward(q2, k2, xt, attn2)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q2, k2, xt, attn2 = ctx.saved_tensors
        B, S, F_, C = xt.shape
        dout = dout.contiguous()
        dq2, dk2, dxt = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(xt)
        _lib.check(_lib.lib().focus_traj_time_bwd(_p(q2), _p(k2), _p(xt), _p(attn2), _p(dout), S * C, _p(dq2), _p(dk2),
                                                  _p(dxt), 0, B, S, F_, ctx.heads, C // ctx.heads, _dt(xt),
                                                  _stream()), "traj_time_bwd")
        return dq2, dk2, dxt, None


def traj_time(q2, k2, xt, heads):
    return _TrajTimeFn.apply(q2, k2, xt, heads)


class _TrajTimeBlockFn(torch.autograd.Function):
    """k2 = proj_kv(x~)[..., :C] and the temporal attention (attention.py:537-549) as ONE autograd node.  x~ feeds
    both the k2 GEMM and the attention, so autograd would sum two [B,S,F,C] gradients (a 460 MB elementwise pass per
    block); here the attention's d(x~) rides into the dX GEMM  d(x~) = dk2 . Wk + d(x~)_attn  as its residual
    epilogue.  The dead v2 half of proj_kv gets its (zero) gradient rows written once instead of through a slice
    backward."""

    @staticmethod
    def forward(ctx, q2, xt, w_kv, b_kv, cls_out, heads):
        _need_gpu(q2, xt, w_kv)
        q2, xt = q2.contiguous(), xt.contiguous()
        B, S, F_, C = xt.shape
        d = C // heads
        wk = shadow(w_kv, xt.dtype)[:C]
        bk = b_kv.detach()[:C] if b_kv is not None else None
        k2 = mm_nt(xt.view(-1, C), wk, bias=bk).view(B, S, F_, C)
        # the attention rows are written straight into tokens 1.. of the [B, 1+S, C] input of `proj`; token 0 is the
        # cls row (attention.py:551 concatenates them)
        out = torch.empty(B, S + 1, C, device=xt.device, dtype=xt.dtype)
        out[:, 0].copy_(cls_out.reshape(B, C))
        attn2 = torch.empty(B, heads, S, F_, device=xt.device, dtype=torch.float32)
        _lib.check(_lib.lib().focus_traj_time_fwd(_p(q2), _p(k2), _p(xt), _p(out, C), (S + 1) * C, _p(attn2), B, S, F_,
                                                  heads, d, _dt(xt), _stream()), "traj_time_fwd")
        ctx.save_for_backward(q2, k2, xt, attn2, w_kv)
        ctx.heads, ctx.has_b, ctx.cls_shape = heads, b_kv is not None, cls_out.shape
        return out

    @staticmethod
    def backward(ctx, dcat):
        q2, k2, xt, attn2, w_kv = ctx.saved_tensors
        B, S, F_, C = xt.shape
        dcat = dcat.contiguous()                 # [B, 1+S, C]: rows 1.. are read in place (batch stride (S+1)*C)
        dq2, dk2, dxt_a = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(xt)
        _lib.check(_lib.lib().focus_traj_time_bwd(_p(q2), _p(k2), _p(xt), _p(attn2), _p(dcat, C), (S + 1) * C, _p(dq2),
                                                  _p(dk2), _p(dxt_a), 0, B, S, F_, ctx.heads, C // ctx.heads, _dt(xt),
                                                  _stream()), "traj_time_bwd")
        dk2f, xt2 = dk2.view(-1, C), xt.view(-1, C)
        dw = db = dwk = dbk = None
        join = None
        if ctx.needs_input_grad[2]:
            dwk, dbk, join = wgrad_async(dk2f, xt2, ctx.has_b and ctx.needs_input_grad[3])
        dxt = None
        if ctx.needs_input_grad[1]:
            if xt.dtype == torch.bfloat16 and C % 64 == 0:
                wt = shadow(w_kv, xt.dtype, transposed=True)            # [C_in, 2C]: columns :C are Wk^T
                dxt = mm_nt(dk2f, wt[:, :C], residual=dxt_a.view(-1, C))
            else:
                dxt = mm_nn(dk2f, shadow(w_kv, xt.dtype)[:C]).add_(dxt_a.view(-1, C))
            dxt = dxt.view(B, S, F_, C)
        if join is not None:
            join()
        if dwk is not None:
            dw = torch.empty(2 * C, C, device=xt.device, dtype=torch.float32)
            dw[:C].copy_(dwk)
            dw[C:].zero_()                      # v2 half: no output use, exactly zero gradient
            if dbk is not None:
                db = torch.zeros(2 * C, device=xt.device, dtype=torch.float32)
                db[:C].copy_(dbk)
        dcls = dcat[:, :1].reshape(ctx.cls_shape) if ctx.needs_input_grad[4] else None
        return dq2, dxt, dw, db, dcls, None


def traj_time_block(q2, xt, w_kv, b_kv, cls_out, heads):
    """[B,1+S,C] input of `proj`: row 0 = cls_out, rows 1.. = the temporal step from q2 [B,S,C], x~ [B,S,F,C] and
    the proj_kv parameters."""
    return _TrajTimeBlockFn.apply(q2, xt, w_kv, b_kv, cls_out, heads)


class _TrajTime2BlockFn(torch.autograd.Function):
    """Temporal step of attention.py:536-549 WITHOUT k2 = proj_kv(x~)[..., :C] in HBM (csrc/traj_time2.hip): the logits
    are scale * (Wk[h]^T q2[s,h,:]) . x~[s,f,:] + const_f, u = Wk[h]^T q2 is formed per 64-channel chunk on chip.
    Forward = 3 HBM-bound launches over x~; backward = 2 launches + two batched GEMMs over the heads from
    g = d(loss)/d(u) [B,S,h,C]:  dq2[:, h] = g[:, h, :] . Wk[h]^T ,  dWk[h] = q2[:, h]^T . g[:, h, :].
    Output: the [B,1+S,C] input of `proj` (row 0 = cls_out), like _TrajTimeBlockFn."""

    @staticmethod
    def forward(ctx, q2, xt, w_kv, b_kv, cls_out, heads):
        _need_gpu(q2, xt, w_kv)
        q2, xt = q2.contiguous(), xt.contiguous()
        B, S, F_, C = xt.shape
        d = C // heads
        L = _lib.lib()
        wkT = shadow(w_kv, xt.dtype, transposed=True)                   # [C_in, 2C]: columns :C are Wk^T
        out = torch.empty(B, S + 1, C, device=xt.device, dtype=xt.dtype)
        out[:, 0].copy_(cls_out.reshape(B, C))
        attn2 = torch.empty(B, S, heads, F_, device=xt.device, dtype=torch.float32)
        nb = L.focus_traj_time2_workspace_bytes(B, S, F_, heads, d)
        ws = torch.empty(nb // 4, device=xt.device, dtype=torch.float32)
        _lib.check(L.focus_traj_time2_fwd(_p(q2), _p(xt), _p(wkT), wkT.stride(0), _p(out, C), (S + 1) * C, _p(attn2),
                                          _p(ws), nb, B, S, F_, heads, d, _dt(xt), _stream()), "traj_time2_fwd")
        ctx.save_for_backward(q2, xt, attn2, w_kv)
        ctx.heads, ctx.has_b, ctx.cls_shape = heads, b_kv is not None, cls_out.shape
        return out

    @staticmethod
    def backward(ctx, dcat):
        q2, xt, attn2, w_kv = ctx.saved_tensors
        heads = ctx.heads
        B, S, F_, C = xt.shape
        d = C // heads
        dev = xt.device
        L = _lib.lib()
        dcat = dcat.contiguous()                 # [B, 1+S, C]: rows 1.. are read in place (batch stride (S+1)*C)
        wkT = shadow(w_kv, xt.dtype, transposed=True)
        dxt = torch.empty_like(xt)
        g = torch.empty(heads, B * S, C, device=dev, dtype=xt.dtype)         # head-major: each head's GEMM operand is dense
        dl = torch.empty(B, S, F_, 16, device=dev, dtype=xt.dtype)           # [.., h padded to 16]
        _lib.check(L.focus_traj_time2_bwd(_p(q2), _p(xt), _p(wkT), wkT.stride(0), _p(attn2), _p(dcat, C), (S + 1) * C,
                                          _p(dxt), _p(g), _p(dl), B, S, F_, heads, d, _dt(xt), _stream()), "traj_time2_bwd")
        R = B * S
        dq2 = dw = db = None
        if (ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and L.focus_traj_time2_gw_ok(R, heads, d, _dt(xt))
                and _os.environ.get("FOCUS_TIME2_GW", "1") != "0"):
            # both consumers of g in one pass over it (csrc/traj_time2_gw.hip): g is 12x the size of dq2
            wk = shadow(w_kv, xt.dtype)                                  # [2C, C] row-major: rows :C are Wk
            dq2 = torch.empty(B, S, C, device=dev, dtype=xt.dtype)
            dw = torch.empty(2 * C, C, device=dev, dtype=torch.float32)
            nbw = L.focus_traj_time2_gw_workspace_bytes(R, heads, d)
            wsw = torch.empty(nbw // 4, device=dev, dtype=torch.float32)
            _lib.check(L.focus_traj_time2_gw(_p(g), _p(q2), _p(wk), wk.stride(0), _p(dq2), _p(dw), _p(wsw), nbw, R, heads, d,
                                             _dt(xt), _stream()), "traj_time2_gw")
            dw[C:].zero_()                      # v2 half: no output use, exactly zero gradient
            if ctx.has_b and ctx.needs_input_grad[3]:
                db = torch.zeros(2 * C, device=dev, dtype=torch.float32)   # softmax over f is shift invariant: exactly 0
            dcls = dcat[:, :1].reshape(ctx.cls_shape) if ctx.needs_input_grad[4] else None
            return dq2, dxt, dw, db, dcls, None
        if ctx.needs_input_grad[0]:
            # dq2[r, h*d+dd] = sum_c g[h,r,c] Wk[h*d+dd, c]   (one launch, batched over the heads)
            wk = shadow(w_kv, xt.dtype)                                  # [2C, C] row-major: rows :C are Wk
            dq2 = torch.empty(B, S, C, device=dev, dtype=xt.dtype)
            gemm(R, d, C, (g, 0), (C, 1, 0, R * C), (wk, 0), (1, C, 0, d * C), (dq2, 0), (C, 1, 0, d),
                 batch=(1, heads))
        if ctx.needs_input_grad[2]:
            # dWk[h*d+dd, c] = sum_r q2[r, h*d+dd] g[h,r,c]   (batched weight-gradient product; slabs summed once)
            dw = torch.empty(2 * C, C, device=dev, dtype=torch.float32)
            nbw = L.focus_gemm_tn_batched_workspace_bytes(d, C, R, heads)
            wsw = torch.empty(max(nbw, 16) // 4, device=dev, dtype=torch.float32)
            gemm(d, C, R, (q2, 0), (1, C, 0, d), (g, 0), (C, 1, 0, R * C), (dw, 0), (C, 1, 0, d * C),
                 batch=(1, heads), aux=(wsw, 0))
            dw[C:].zero_()                      # v2 half: no output use, exactly zero gradient
            if ctx.has_b and ctx.needs_input_grad[3]:
                db = torch.zeros(2 * C, device=dev, dtype=torch.float32)   # softmax over f is shift invariant: exactly 0
        dcls = dcat[:, :1].reshape(ctx.cls_shape) if ctx.needs_input_grad[4] else None
        return dq2, dxt, dw, db, dcls, None


def traj_time2_ok(xt, heads):
    """The k2-free temporal step takes bf16, head dim 64, <= 16 heads, F in {4, 8, 16} (FOCUS_TIME2=0 disables it)."""
    if _os.environ.get("FOCUS_TIME2", "1") == "0" or xt.dtype != torch.bfloat16:
        return False
    B, S, F_, C = xt.shape
    return C % heads == 0 and C // heads == 64 and heads <= 16 and F_ in (4, 8, 16)


def traj_time2_block(q2, xt, w_kv, b_kv, cls_out, heads):
    """[B,1+S,C] input of `proj` (row 0 = cls_out) from q2 [B,S,C], x~ [B,S,F,C] and the proj_kv parameters."""
    return _TrajTime2BlockFn.apply(q2, xt, w_kv, b_kv, cls_out, heads)


# --------------------------------------------------------------------------------------------------
# Small joint attention (motion stream: attention.py:369-385; slot predictor: transformer.py:23-49)
# --------------------------------------------------------------------------------------------------
class _SmallAttnFn(torch.autograd.Function):
    """softmax(scale * q k^T [causal]) [* drop] v per head as strided batched GEMMs + a row softmax; q [B,Nq,C],
    k, v [B,Nk,C] (may be views of one buffer).  `drop` [B,h,Nq,Nk] is an inverted-dropout mask (already scaled by
    1/(1-p)) applied to the probabilities (transformer.py:45), or None."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, causal, drop):
        _need_gpu(q, k, v)
        B, N, C = q.shape
        M = k.shape[1]
        d = C // heads
        dev, dt = q.device, q.dtype
        att = torch.empty(B, heads, N, M, device=dev, dtype=dt)
        L = _lib.lib()
        ctx.heads, ctx.scale = heads, scale

        def rows_ok(t):      # [B, n, C] whose rows are ld apart with no gap between clips (dense, or a column block)
            return t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) >= C

        ctx.fused = (not causal and drop is None and bool(L.focus_small_attn_ok(N, M, d))
                     and rows_ok(q) and rows_ok(k) and rows_ok(v))
        if ctx.fused:       # a handful of tokens (the slot predictor): the whole attention in one launch
            out = torch.empty(B, N, C, device=dev, dtype=dt)
            _lib.check(L.focus_small_attn_fwd(_p(q), _p(k), _p(v), q.stride(1), k.stride(1), v.stride(1), _p(att), _p(out), B,
                                              heads, N, M, d, scale, _dt(q), _stream()), "small_attn_fwd")
            ctx.save_for_backward(q, k, v, att, None)
            return out
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        sa = (M, 1, heads * N * M, N * M)
        gemm(N, M, d, (q, 0), (C, 1, N * C, d), (k, 0), (1, C, M * C, d), (att, 0), sa, batch=(B, heads))
        if causal:
            assert N == M, "causal self-attention"
            _lib.check(L.focus_softmax_causal_fwd(_p(att), _p(att), B * heads * N, M, M, N, scale, _dt(att), _stream()),
                       "softmax_causal_fwd")
        else:
            _lib.check(L.focus_softmax_fwd(_p(att), _p(att), B * heads * N, M, M, scale, _dt(att), _stream()),
                       "softmax_fwd")
        attd = att if drop is None else att * drop
        out = torch.empty(B, N, C, device=dev, dtype=dt)
        gemm(N, d, M, (attd, 0), sa, (v, 0), (C, 1, M * C, d), (out, 0), (C, 1, N * C, d), batch=(B, heads))
        ctx.save_for_backward(q, k, v, att, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, att, drop = ctx.saved_tensors
        heads, scale = ctx.heads, ctx.scale
        B, N, C = q.shape
        M = k.shape[1]
        d = C // heads
        dout = dout.contiguous()
        if ctx.fused:
            es = q.element_size()
            if (N == M and q.stride(1) == 3 * C == k.stride(1) == v.stride(1) and k.data_ptr() == q.data_ptr() + C * es
                    and v.data_ptr() == k.data_ptr() + C * es):
                # q | k | v are the column blocks of one projection output (ops.linear_qkv): so are their gradients
                dqkv = torch.empty(B, N, 3 * C, device=q.device, dtype=q.dtype)
                dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
            else:
                dq = torch.empty(q.shape, device=q.device, dtype=q.dtype)
                dk = torch.empty(k.shape, device=q.device, dtype=q.dtype)
                dv = torch.empty(v.shape, device=q.device, dtype=q.dtype)
                if not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
                    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
            _lib.check(_lib.lib().focus_small_attn_bwd(_p(q), _p(k), _p(v), q.stride(1), k.stride(1), v.stride(1), _p(att),
                                                       _p(dout), _p(dq), _p(dk), _p(dv), B, heads, N, M, d, scale, _dt(q),
                                                       _stream()), "small_attn_bwd")
            return dq, dk, dv, None, None, None, None
        sa = (M, 1, heads * N * M, N * M)
        sat = (1, M, heads * N * M, N * M)
        sq = (C, 1, N * C, d)
        sk = (C, 1, M * C, d)
        attd = att if drop is None else att * drop
        datt = torch.empty_like(att)
        # dA = dout . v^T ; dV = A^T . dout
        gemm(N, M, d, (dout, 0), sq, (v, 0), (1, C, M * C, d), (datt, 0), sa, batch=(B, heads))
        dv = torch.empty_like(v)
        gemm(M, d, N, (attd, 0), sat, (dout, 0), sq, (dv, 0), sk, batch=(B, heads))
        if drop is not None:
            datt = datt * drop
        _lib.check(_lib.lib().focus_softmax_bwd(_p(datt), _p(att), _p(datt), B * heads * N, M, M, scale, _dt(att),
                                                _stream()), "softmax_bwd")
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        gemm(N, d, M, (datt, 0), sa, (k, 0), sk, (dq, 0), sq, batch=(B, heads))
        gemm(M, d, N, (datt, 0), sat, (q, 0), sq, (dk, 0), sk, batch=(B, heads))
        return dq, dk, dv, None, None, None, None


# --------------------------------------------------------------------------------------------------
# Flash attention (STEVE decoder: causal self-attention over the image tokens, transformer.py:23-49, :131-151)
# --------------------------------------------------------------------------------------------------
def flash_ok(q, k, v, heads, causal):
    """bf16 [B, n, C] row views (dense or column blocks of one projection output) with head dim 32 / 48 / 64."""
    if not (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == q.dtype and v.dtype == q.dtype and q.dim() == 3):
        return False
    C = q.shape[2]
    if C % heads:
        return False
    for t in (q, k, v):
        if t.stride(2) != 1 or t.stride(1) % 8 or t.stride(0) % 8 or t.data_ptr() % 16:
            return False
    return bool(_lib.lib().focus_flash_attn_ok(q.shape[1], k.shape[1], C // heads, _lib.BF16, int(bool(causal))))


def drop_threshold(p):
    """The 16-bit threshold the kernels compare against: p is quantised to thr / 65536 (0.1 -> 6554)."""
    thr = int(round(float(p) * 65536.0))
    assert 0 <= thr < 65536, "dropout probability out of range"
    return thr


def _flash_args(q, k, v, out, lse, heads, scale, causal, thr, seed):
    B, Nq, C = q.shape
    a = _lib.FlashArgs()
    a.q, a.k, a.v, a.out, a.lse = _p(q), _p(k), _p(v), _p(out), _p(lse)
    a.seed = _p(seed) if seed is not None else None
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    a.bsq, a.bsk, a.bsv, a.bso = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.B, a.heads, a.Nq, a.Nk, a.d, a.dtype, a.causal = B, heads, Nq, k.shape[1], C // heads, _lib.BF16, int(bool(causal))
    a.drop_thr, a.scale = thr, scale
    return a


class _FlashAttnFn(torch.autograd.Function):
    """out = dropout(softmax(scale q k^T [+ causal mask])) v per head, nothing of size Nq x Nk in memory
    (csrc/flash_attn.hip).  `seed`: int32 [1] device tensor naming the dropout draw (None: no dropout)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, causal, thr, seed):
        _need_gpu(q, k, v)
        B, Nq, C = q.shape
        out = torch.empty(B, Nq, C, device=q.device, dtype=q.dtype)
        lse = torch.empty(B, heads, Nq, device=q.device, dtype=torch.float32)
        a = _flash_args(q, k, v, out, lse, heads, scale, causal, thr, seed)
        _lib.check(_lib.lib().focus_flash_attn_fwd(ctypes.byref(a), _stream()), "flash_attn_fwd")
        ctx.save_for_backward(q, k, v, out, lse, seed)
        ctx.cfg = (heads, scale, causal, thr)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, seed = ctx.saved_tensors
        heads, scale, causal, thr = ctx.cfg
        B, Nq, C = q.shape
        Nk = k.shape[1]
        dout = dout.contiguous()
        es = q.element_size()
        if (Nq == Nk and q.stride(1) == 3 * C == k.stride(1) == v.stride(1) and k.data_ptr() == q.data_ptr() + C * es
                and v.data_ptr() == k.data_ptr() + C * es):
            # q | k | v are the column blocks of one projection output (ops.linear_qkv): so are their gradients
            dqkv = torch.empty(B, Nq, 3 * C, device=q.device, dtype=q.dtype)
            dq, dk, dv = dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:]
        else:
            dq = torch.empty(B, Nq, C, device=q.device, dtype=q.dtype)
            dk = torch.empty(B, Nk, C, device=q.device, dtype=q.dtype)
            dv = torch.empty(B, Nk, C, device=q.device, dtype=q.dtype)
        delta = torch.empty(B, heads, Nq, device=q.device, dtype=torch.float32)
        a = _flash_args(q, k, v, out, lse, heads, scale, causal, thr, seed)
        a.dout, a.delta, a.dq, a.dk, a.dv = _p(dout), _p(delta), _p(dq), _p(dk), _p(dv)
        a.lddo, a.lddq, a.lddk, a.lddv = dout.stride(1), dq.stride(1), dk.stride(1), dv.stride(1)
        a.bsdo, a.bsdq, a.bsdk, a.bsdv = dout.stride(0), dq.stride(0), dk.stride(0), dv.stride(0)
        _lib.check(_lib.lib().focus_flash_attn_bwd(ctypes.byref(a), _stream()), "flash_attn_bwd")
        return dq, dk, dv, None, None, None, None, None


def flash_attention(q, k, v, heads, scale, causal=False, p=0.0, seed=None):
    """Attention over [B, n, C] rows without materialised probabilities.  p > 0: dropout on the probabilities
    (transformer.py:44-45) drawn from `seed` (int32 [1] on the device; a fresh one from torch's generator when None)."""
    thr = drop_threshold(p) if p > 0.0 else 0
    if thr and seed is None:
        seed = torch.randint(0, 2 ** 31 - 1, (1,), device=q.device, dtype=torch.int32)
    return _FlashAttnFn.apply(q, k, v, heads, scale, causal, thr, seed if thr else None)


def small_attention(q, k, v, heads, scale, causal=False, drop=None):
    return _SmallAttnFn.apply(q, k, v, heads, scale, causal, drop)


# --------------------------------------------------------------------------------------------------
# RoIAlign, per-RoI max, box layout
# --------------------------------------------------------------------------------------------------
class _RoiAlignFn(torch.autograd.Function):
    """RoIAlign over channels-last token maps.  `feat` is either a dense map [NI, H*W, C] (tok_T == 0) or the residual
    stream itself [B, 1 + T*H*W, C] (tok_T == T): the maps of image (b, t) are then read in place behind the cls row,
    and the backward returns a gradient of the stream's shape (cls rows zero) -- no slice copy in, no slice_backward
    + zero-fill out.  rois [K,4] fp32 xyxy pixels."""

    @staticmethod
    def forward(ctx, feat, rois, roi_img, tok_T, H, W, PH, PW, scale, sampling_ratio, aligned, relu):
        _need_gpu(feat, rois, roi_img)
        feat = feat.contiguous()
        C = feat.shape[-1]
        if tok_T:
            B = feat.shape[0]
            assert feat.shape[1] == 1 + tok_T * H * W
            NI, ipb, bstride, off = B * tok_T, tok_T, (1 + tok_T * H * W) * C, C
        else:
            NI, HW, _ = feat.shape
            assert HW == H * W
            ipb, bstride, off = NI, NI * H * W * C, 0
        K = rois.shape[0]
        out = torch.empty(K, PH * PW, C, device=feat.device, dtype=feat.dtype)
        _lib.check(_lib.lib().focus_roi_align_fwd(_p(feat, off), H * W * C, ipb, bstride, _p(rois), _p(roi_img), _p(out), NI,
                                                  C, H, W, K, PH, PW, scale, sampling_ratio, int(aligned), int(relu),
                                                  _dt(feat), _stream()), "roi_align_fwd")
        ctx.relu = bool(relu)
        ctx.save_for_backward(rois, roi_img, out if relu else None)
        ctx.args = (NI, C, H, W, K, PH, PW, scale, sampling_ratio, int(aligned), feat.dtype, ipb, bstride, off, feat.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        rois, roi_img, relu_out = ctx.saved_tensors
        NI, C, H, W, K, PH, PW, scale, sr, al, dt, ipb, bstride, off, shp = ctx.args
        dout = dout.contiguous()
        L = _lib.lib()
        dfeat = torch.empty(shp, device=dout.device, dtype=dt)
        if off:
            dfeat[:, 0].zero_()                                   # cls rows: no RoI reads them
        nb = L.focus_roi_align_bwd_workspace_bytes(NI, C, H, W, PH, PW)
        ws = torch.empty(nb // 4, device=dout.device, dtype=torch.float32) if nb else None
        _lib.check(L.focus_roi_align_bwd(_p(dout), _p(relu_out), _p(rois), _p(roi_img), _p(dfeat, off), H * W * C, ipb, bstride,
                                         _p(ws), nb, NI, C, H, W, K, PH, PW, scale, sr, al, _dt(dout), _stream()),
                   "roi_align_bwd")
        return dfeat, None, None, None, None, None, None, None, None, None, None, None


def roi_align_tokens(feat, rois, roi_img, H, W, PH, PW, spatial_scale, sampling_ratio=-1, aligned=True):
    return _RoiAlignFn.apply(feat, rois, roi_img, 0, H, W, PH, PW, float(spatial_scale), sampling_ratio, aligned, False)


def roi_align_stream(x, rois, roi_img, T, H, W, PH, PW, spatial_scale, sampling_ratio=-1, aligned=True, relu=False):
    """RoIAlign of the patch tokens of the residual stream x [B, 1+T*H*W, C], read in place (image index = b*T + t).
    Shapes whose backward needs the dense atomic path (focus_roi_align_bwd_workspace_bytes > 0) take a dense copy.
    relu: max(., 0) fused into the forward store and (as a mask from the saved output) into the backward."""
    B, _, C = x.shape
    if _lib.lib().focus_roi_align_bwd_workspace_bytes(B * T, C, H, W, PH, PW):
        out = roi_align_tokens(x[:, 1:].reshape(B * T, H * W, C), rois, roi_img, H, W, PH, PW, spatial_scale,
                               sampling_ratio, aligned)
        return torch.relu(out) if relu else out
    return _RoiAlignFn.apply(x, rois, roi_img, T, H, W, PH, PW, float(spatial_scale), sampling_ratio, aligned, relu)


def roi_stream_supports_fused(x, T, H, W, PH, PW):
    """True when roi_align_stream takes the in-place separable path for this shape (then its ReLU is fused too)."""
    B, _, C = x.shape
    return _lib.lib().focus_roi_align_bwd_workspace_bytes(B * T, C, H, W, PH, PW) == 0


class _OrvitAssembleFn(torch.autograd.Function):
    """all = cat(cls, per frame [patch tokens, object tokens]) (orvit.py:145-147) and its adjoint, one pass each."""

    @staticmethod
    def forward(ctx, x, obj, T, HW):
        _need_gpu(x, obj)
        x, obj = x.contiguous(), obj.contiguous()
        B, _, C = x.shape
        O = obj.shape[2]
        assert x.shape[1] == 1 + T * HW and obj.shape[:2] == (B, T) and obj.dtype == x.dtype
        out = torch.empty(B, 1 + T * (HW + O), C, device=x.device, dtype=x.dtype)
        _lib.check(_lib.lib().focus_orvit_assemble(_p(x), _p(obj), _p(out), B, T, HW, O, C, _dt(x), _stream()),
                   "orvit_assemble")
        ctx.args = (B, T, HW, O, C)
        return out

    @staticmethod
    def backward(ctx, dall):
        B, T, HW, O, C = ctx.args
        dall = dall.contiguous()
        dx = torch.empty(B, 1 + T * HW, C, device=dall.device, dtype=dall.dtype)
        dobj = torch.empty(B, T, O, C, device=dall.device, dtype=dall.dtype)
        _lib.check(_lib.lib().focus_orvit_assemble_bwd(_p(dall), _p(dx), _p(dobj), B, T, HW, O, C, _dt(dall), _stream()),
                   "orvit_assemble_bwd")
        return dx, dobj, None, None


def orvit_assemble(x, obj, T, HW):
    """x [B,1+T*HW,C], obj [B,T,O,C] -> [B, 1+T*(HW+O), C]."""
    return _OrvitAssembleFn.apply(x, obj, T, HW)


class _OrvitMergeFn(torch.autograd.Function):
    """x + s_b * (patch/cls rows of y + [0; mm]) (orvit.py:152-169: slice, reshape copy, motion residual, cat, drop-path,
    residual add) as one pass; the adjoint writes dy (object rows zero) and dmm, dx is dout itself."""

    @staticmethod
    def forward(ctx, x, y, mm, draws, keep, T, HW):
        _need_gpu(x, y)
        x, y = x.contiguous(), y.contiguous()
        B, _, C = x.shape
        O = (y.shape[1] - 1) // T - HW
        assert y.shape[1] == 1 + T * (HW + O) and x.shape[1] == 1 + T * HW
        scale = None
        if draws is not None:                                     # stochastic depth: s_b = floor(keep + u_b) / keep
            scale = torch.floor(keep + draws) / keep
        if mm is not None:
            mm = mm.contiguous()
        out = torch.empty_like(x)
        _lib.check(_lib.lib().focus_orvit_merge(_p(x), _p(y), _p(mm) if mm is not None else None,
                                                _p(scale) if scale is not None else None, _p(out), B, T, HW, O, C, _dt(x),
                                                _stream()), "orvit_merge")
        ctx.scale = scale
        ctx.args = (B, T, HW, O, C, mm is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, HW, O, C, has_mm = ctx.args
        dout = dout.contiguous()
        dy = torch.empty(B, 1 + T * (HW + O), C, device=dout.device, dtype=dout.dtype)
        dmm = torch.empty(B, T * HW, C, device=dout.device, dtype=dout.dtype) if has_mm else None
        sc = ctx.scale
        _lib.check(_lib.lib().focus_orvit_merge_bwd(_p(dout), _p(sc) if sc is not None else None, _p(dy),
                                                    _p(dmm) if has_mm else None, B, T, HW, O, C, _dt(dout), _stream()),
                   "orvit_merge_bwd")
        return dout, dy, dmm, None, None, None, None


def orvit_merge(x, y, mm, T, HW, drop_prob=0.0, training=False):
    """x [B,1+T*HW,C] + drop_path(cat(y[:, :1], patch rows of y + mm)); y [B,1+T*(HW+O),C], mm [B,T*HW,C] or None."""
    draws = None
    keep = 1.0 - drop_prob
    if drop_prob > 0.0 and training:
        draws = torch.rand(x.shape[0], dtype=torch.float32, device=x.device)
    return _OrvitMergeFn.apply(x, y, mm, draws, keep, T, HW)


def roi_align_indices(rois, H, W, PH, PW, spatial_scale, sampling_ratio=-1, aligned=True):
    """Integer side of RoIAlign (parity export): grid [K,2], neighbours [K,PH,PW,4] int32."""
    _need_gpu(rois)
    K = rois.shape[0]
    grid = torch.empty(K, 2, device=rois.device, dtype=torch.int32)
    nbr = torch.empty(K, PH, PW, 4, device=rois.device, dtype=torch.int32)
    _lib.check(_lib.lib().focus_roi_align_indices(_p(rois), _p(grid), _p(nbr), H, W, K, PH, PW, float(spatial_scale),
                                                  sampling_ratio, int(aligned), _stream()), "roi_align_indices")
    return grid, nbr


class _CellAmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        x = x.contiguous()
        K, cells, C = x.shape
        y = torch.empty(K, C, device=x.device, dtype=x.dtype)
        arg = torch.empty(K, C, device=x.device, dtype=torch.int32)
        _lib.check(_lib.lib().focus_cell_amax_fwd(_p(x), _p(y), _p(arg), K, cells, C, _dt(x), _stream()), "cell_amax")
        ctx.save_for_backward(arg)
        ctx.cells = cells
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        K, C = arg.shape
        dy = dy.contiguous()
        dx = torch.empty(K, ctx.cells, C, device=dy.device, dtype=dy.dtype)
        _lib.check(_lib.lib().focus_cell_amax_bwd(_p(dy), _p(arg), _p(dx), K, ctx.cells, C, _dt(dy), _stream()),
                   "cell_amax_bwd")
        return dx


def cell_amax(x):
    """max over the cells of each RoI: [K, cells, C] -> [K, C]."""
    return _CellAmaxFn.apply(x)


class _BoxLayoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vecs, boxes, H, W):
        _need_gpu(vecs, boxes)
        vecs, boxes = vecs.contiguous(), boxes.contiguous().float()
        NF, O, C = vecs.shape
        out = torch.empty(NF, H * W, C, device=vecs.device, dtype=vecs.dtype)
        _lib.check(_lib.lib().focus_box_layout_fwd(_p(vecs), _p(boxes), _p(out), NF, O, C, H, W, _dt(vecs), _stream()),
                   "box_layout_fwd")
        ctx.save_for_backward(boxes)
        ctx.args = (NF, O, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        (boxes,) = ctx.saved_tensors
        NF, O, C, H, W = ctx.args
        dout = dout.contiguous()
        dv = torch.empty(NF, O, C, device=dout.device, dtype=dout.dtype)
        _lib.check(_lib.lib().focus_box_layout_bwd(_p(dout), _p(boxes), _p(dv), NF, O, C, H, W, _dt(dout), _stream()),
                   "box_layout_bwd")
        return dv, None, None, None


def box_layout(vecs, boxes, H, W):
    """vecs [NF,O,C], boxes [NF,O,4] cxcywh -> [NF, H*W, C]."""
    return _BoxLayoutFn.apply(vecs, boxes, H, W)


# --------------------------------------------------------------------------------------------------
# Motionformer embedding / loss
# --------------------------------------------------------------------------------------------------
def im2col_patches(x, kt, kh, kw, dtype):
    _need_gpu(x)
    x = x.contiguous().float()
    B, Cin, T, H, W = x.shape
    rows = B * (T // kt) * (H // kh) * (W // kw)
    cols = torch.empty(rows, Cin * kt * kh * kw, device=x.device, dtype=dtype)
    _lib.check(_lib.lib().focus_im2col_patches(_p(x), _p(cols), B, Cin, T, H, W, kt, kh, kw, _dt(cols), _stream()),
               "im2col")
    return cols


class _EmbedAssembleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patch, cls, pos, temp):
        _need_gpu(patch, cls, pos, temp)
        B, S, C = patch.shape
        T = temp.shape[0]
        P = S // T
        patch = patch.contiguous()
        tok = torch.empty(B, 1 + S, C, device=patch.device, dtype=patch.dtype)
        _lib.check(_lib.lib().focus_embed_assemble(_p(patch), _p(cls), _p(pos), _p(temp), _p(tok), B, T, P, C,
                                                   _dt(patch), _stream()), "embed_assemble")
        ctx.dims = (B, T, P, C)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        # adjoint = slices and small reductions over the batch: parameter-gradient plumbing
        B, T, P, C = ctx.dims
        g = dtok.float()
        gp = g[:, 1:].reshape(B, T, P, C)
        dcls = g[:, 0].sum(0)
        dpos = torch.cat([dcls[None], gp.sum((0, 1))], 0)
        dtemp = gp.sum((0, 2))
        return dtok[:, 1:], dcls, dpos, dtemp


def embed_assemble(patch, cls, pos, temp):
    """cls [C], pos [1+P,C], temp [T,C] fp32 parameters."""
    return _EmbedAssembleFn.apply(patch, cls.contiguous(), pos.contiguous(), temp.contiguous())


def rows_ok(t):
    """[R, V] rows the one-workgroup-per-row passes of gumbel.hip take: dense, fp32 or bf16, V % 8 == 0, V <= 8192."""
    return (t.is_cuda and t.dim() == 2 and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
            and t.data_ptr() % 16 == 0 and bool(_lib.lib().focus_rows_ok(t.shape[0], t.shape[1], _dt(t))))


class _XentLsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smoothing):
        _need_gpu(logits, target)
        ctx.rows = rows_ok(logits)
        R, Ncls = logits.shape
        loss_rows = torch.empty(R, device=logits.device, dtype=torch.float32)
        if ctx.rows:
            # fp32 or bf16 logits as they are; the backward rebuilds softmax from the row log-sum-exp instead of keeping an
            # fp32 [R, V] gradient alive (STEVE's decoder head: 196608 x 4096 per 8 clips)
            target = target.contiguous()
            lse = torch.empty(R, device=logits.device, dtype=torch.float32)
            _lib.check(_lib.lib().focus_xent_rows_fwd(_p(logits), _p(target), _p(loss_rows), _p(lse), R, Ncls, smoothing,
                                                      _dt(logits), _stream()), "xent_rows_fwd")
            ctx.save_for_backward(logits, target, lse)
            ctx.smoothing = smoothing
            return loss_rows.mean()
        logits = logits.float().contiguous()
        dlog = torch.empty_like(logits)
        _lib.check(_lib.lib().focus_xent_ls(_p(logits), _p(target.contiguous()), _p(loss_rows), _p(dlog), R, Ncls,
                                            smoothing, _stream()), "xent_ls")
        ctx.save_for_backward(dlog)
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, g):
        if ctx.rows:
            logits, target, lse = ctx.saved_tensors
            dlog = torch.empty_like(logits)
            g = g.to(torch.float32).contiguous()
            _lib.check(_lib.lib().focus_xent_rows_bwd(_p(logits), _p(target), _p(lse), _p(g), _p(dlog), logits.shape[0],
                                                      logits.shape[1], ctx.smoothing, _dt(logits), _stream()), "xent_rows_bwd")
            return dlog, None, None
        (dlog,) = ctx.saved_tensors
        return dlog * g, None, None


def label_smoothing_ce(logits, target, smoothing=0.1):
    return _XentLsFn.apply(logits, target, smoothing)


# --------------------------------------------------------------------------------------------------
# residual + dropout(branch) (the decoder blocks of STEVE: transformer.py:45-47, :147-163)
# --------------------------------------------------------------------------------------------------
class _DropoutAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, residual, thr, seed):
        out = torch.empty_like(y)
        _lib.check(_lib.lib().focus_dropout_add(_p(y), _p(residual) if residual is not None else None, _p(seed), thr, _p(out),
                                                y.numel(), _dt(y), _stream()), "dropout_add")
        ctx.save_for_backward(seed)
        ctx.thr, ctx.has_res = thr, residual is not None
        return out

    @staticmethod
    def backward(ctx, g):
        (seed,) = ctx.saved_tensors
        g = g.contiguous()
        dy = torch.empty_like(g)
        _lib.check(_lib.lib().focus_dropout_add(_p(g), None, _p(seed), ctx.thr, _p(dy), g.numel(), _dt(g), _stream()),
                   "dropout_add_bwd")
        return dy, (g if ctx.has_res else None), None, None


def dropout_add(y, residual, p, training, seed=None):
    """residual + F.dropout(y, p, training) (residual may be None).  On the GPU, for dense fp32 / bf16 tensors, one pass whose
    mask is a hash of (seed, element index) -- rebuilt by the backward, never stored; `seed`: int32 [2] on the device
    (default: drawn from torch's generator).  p is quantised to drop_threshold(p) / 65536."""
    if not training or p <= 0.0:
        return y if residual is None else residual + y
    _need_gpu(y, residual)
    thr = drop_threshold(p)
    # (shapes the kernel does not take -- odd element counts, strided views -- keep ATen's dropout, on the GPU)
    ok = (y.is_contiguous() and y.dtype in (torch.float32, torch.bfloat16) and y.numel() % 8 == 0
          and y.numel() > 0 and y.data_ptr() % 16 == 0 and thr > 0
          and (residual is None or (residual.shape == y.shape and residual.dtype == y.dtype and residual.is_contiguous()
                                    and residual.data_ptr() % 16 == 0)))
    if not ok:
        out = torch.nn.functional.dropout(y, p, True)
        return out if residual is None else residual + out
    if seed is None:
        seed = torch.randint(-2 ** 31, 2 ** 31 - 1, (2,), device=y.device, dtype=torch.int32)
    return _DropoutAddFn.apply(y, residual, thr, seed)


# --------------------------------------------------------------------------------------------------
# Gumbel-softmax over the dVAE vocabulary (steve.py:262-271, STEVE/utils.py:47-61)
# --------------------------------------------------------------------------------------------------
class _GumbelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tau, hard, e_soft, e_hard, want_target, seed, out_dtype):
        _need_gpu(x)
        R, V = x.shape
        dev = x.device
        gen = e_soft is None
        if gen:
            # the draws are made in the kernel from this seed (and made again by the backward); the seed itself comes from
            # torch's generator, on the device, so that a captured step draws fresh noise on every replay
            if seed is None:
                seed = torch.randint(-2 ** 31, 2 ** 31 - 1, (4,), device=dev, dtype=torch.int32)
            assert seed.dtype == torch.int32 and seed.numel() == 4 and seed.is_cuda and seed.is_contiguous()
            e_hard = None
        else:
            e_soft = e_soft.to(torch.float32).contiguous()
            e_hard = e_hard.to(torch.float32).contiguous() if want_target else None
        z = torch.empty_like(x, dtype=out_dtype or x.dtype)
        target = torch.empty(R, device=dev, dtype=torch.int64) if want_target else None
        stats = torch.empty(R, 4, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().focus_gumbel_fwd(_p(x), _p(e_soft) if e_soft is not None else None,
                                               _p(e_hard) if e_hard is not None else None,
                                               _p(seed) if gen else None, _p(z),
                                               _p(target) if target is not None else None, _p(stats), R, V, float(tau),
                                               int(bool(hard)), _dt(x), _dt(z), _stream()), "gumbel_fwd")
        ctx.save_for_backward(x, stats, *([seed] if gen else [e_soft]))
        ctx.gen, ctx.tau, ctx.zdtype = gen, float(tau), z.dtype
        if target is None:
            return z, None
        ctx.mark_non_differentiable(target)
        return z, target

    @staticmethod
    def backward(ctx, dz, _dtarget):
        x, stats, noise = ctx.saved_tensors
        dz = dz.contiguous()
        if dz.dtype != ctx.zdtype:
            dz = dz.to(ctx.zdtype)
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().focus_gumbel_bwd(_p(x), None if ctx.gen else _p(noise), _p(noise) if ctx.gen else None,
                                               _p(stats), _p(dz), _p(dx), x.shape[0], x.shape[1], ctx.tau, _dt(x), _dt(dz), _stream()),
                   "gumbel_bwd")
        return dx, None, None, None, None, None, None, None


def gumbel_softmax_rows(x, tau, hard, e_soft=None, e_hard=None, want_target=True, seed=None, out_dtype=None):
    """x [R, V] raw logits (rows_ok) -> (z [R, V] in x's dtype -- or bf16 from fp32 logits with out_dtype --, target [R] int64
    or None).

    z = gumbel_softmax(log_softmax(x), tau, hard) of STEVE/utils.py:47-61; target = the arg-max of the reference's second,
    hard sample (steve.py:268-269), which is the arg-max of log_softmax(x) + G_hard.  e_soft / e_hard: the Exp(1) draws
    (both given, e.g. the reference's own in a parity test) or None: drawn in the kernel, regenerated by the backward, from
    `seed` (int32 [4] on the device: two words per stream; default: drawn from torch's generator)."""
    assert (e_soft is None) == (e_hard is None) or not want_target, "pass both noise tensors or neither"
    return _GumbelFn.apply(x, tau, hard, e_soft, e_hard, want_target, seed, out_dtype)



# --------------------------------------------------------------------------------------------------
# Slot attention
# --------------------------------------------------------------------------------------------------
class SlotKVGrad:
    """Shared gradient state for the k_t / v_t of ONE frame.  The corrector iterations of a frame all read the same k_t,
    v_t (steve.py:68-83); autograd would sum their three 50 MB gradients with separate add kernels (96 adds per BASELINE
    step).  Iterations that carry the same SlotKVGrad hand autograd ONE gradient, from the node whose backward runs
    last (the others return None) -- valid because every iteration of a frame is on the path to the loss (the slots
    chain through them).  bf16 / K <= 16 / <= 4 iterations: the per-iteration backward only writes its 64-byte
    (w, dlogits) rows and the last node forms dk, dv for all iterations with one MFMA kernel (focus_slot_kv_grad);
    otherwise the backward kernels add into one buffer (focus_slot_attn_bwd accumulate=1)."""

    def __init__(self):
        self.dk = self.dv = None
        self.pending = 0
        self.total = 0
        self.items = []              # deferred mode: (wl, q, dupd) per iteration


class _SlotAttnFn(torch.autograd.Function):
    """One corrector iteration of steve.py:76-83 on one frame.  k_t, v_t [B,N,D]; q [B,K,D]."""

    @staticmethod
    def forward(ctx, k_t, v_t, q, eps, acc):
        _need_gpu(k_t, v_t, q)
        k_t, v_t, q = k_t.contiguous(), v_t.contiguous(), q.contiguous()
        B, N, D = k_t.shape
        K = q.shape[1]
        dev, dt = k_t.device, k_t.dtype
        attn = torch.empty(B, N, K, device=dev, dtype=dt)
        upd = torch.empty(B, K, D, device=dev, dtype=dt)
        cs = torch.empty(B, K, device=dev, dtype=torch.float32)
        L = _lib.lib()
        nb = L.focus_slot_attn_workspace_bytes(B, N, K, D)
        ws = torch.empty(nb, device=dev, dtype=torch.uint8)
        _lib.check(L.focus_slot_attn_fwd(_p(k_t), _p(v_t), N * D, _p(q), _p(attn), N * K, _p(upd), _p(cs), _p(ws), nb,
                                         B, N, K, D, eps, _dt(k_t), _stream()), "slot_attn_fwd")
        ctx.save_for_backward(k_t, v_t, q, attn, cs, upd)
        ctx.eps = eps
        ctx.acc = acc
        if acc is not None:
            acc.pending += 1
            acc.total += 1
        return upd, attn

    @staticmethod
    def backward(ctx, dupd, dattn):
        k_t, v_t, q, attn, cs, upd = ctx.saved_tensors
        B, N, D = k_t.shape
        K = q.shape[1]
        dupd = dupd.contiguous()
        dattn = dattn.contiguous() if dattn is not None else None
        acc = ctx.acc
        L = _lib.lib()
        nb = L.focus_slot_attn_workspace_bytes(B, N, K, D)
        ws = torch.empty(nb, device=k_t.device, dtype=torch.uint8)
        dq = torch.empty_like(q)
        if acc is not None:
            _guard_shared(acc, "slot_attn_step")
        if acc is not None and L.focus_slot_kv_grad_ok(K, D, _dt(k_t), acc.total):
            # deferred: this launch writes its (w, dlogits) rows; the frame's last node forms dk, dv for all iterations
            wl = torch.empty(B, N, 32, device=k_t.device, dtype=k_t.dtype)
            _lib.check(L.focus_slot_attn_bwd(_p(k_t), _p(v_t), N * D, _p(q), _p(attn), N * K, _p(cs), _p(upd), _p(dupd),
                                             _p(dattn), None, None, 0, _p(dq), _p(ws), nb, B, N, K, D, ctx.eps,
                                             _dt(k_t), _p(wl), _stream()), "slot_attn_bwd")
            acc.items.append((wl, q, dupd))
            acc.pending -= 1
            if acc.pending > 0:
                return None, None, dq, None, None
            it = acc.items + [(None, None, None)] * (4 - len(acc.items))
            # one [B,N,2D] matrix [dk | dv]: the projections' backward (linear_kv) then runs one product per gradient
            dkv = torch.empty(B, N, 2 * D, device=k_t.device, dtype=k_t.dtype)
            dk, dv = dkv[..., :D], dkv[..., D:]
            _lib.check(L.focus_slot_kv_grad(*[_p(e[0]) for e in it], *[_p(e[1]) for e in it], *[_p(e[2]) for e in it],
                                            len(acc.items), _p(dk), _p(dv), N * 2 * D, 2 * D, B, N, K, D, _dt(k_t),
                                            _stream()), "slot_kv_grad")
            acc.items = []
            acc.total = 0
            return dk, dv, dq, None, None
        accumulate = 0
        if acc is not None and acc.dk is not None:
            dk, dv, accumulate = acc.dk, acc.dv, 1
        else:
            dk, dv = torch.empty_like(k_t), torch.empty_like(v_t)
            if acc is not None:
                acc.dk, acc.dv = dk, dv
        _lib.check(L.focus_slot_attn_bwd(_p(k_t), _p(v_t), N * D, _p(q), _p(attn), N * K, _p(cs), _p(upd), _p(dupd),
                                         _p(dattn), _p(dk), _p(dv), accumulate, _p(dq), _p(ws), nb, B, N, K, D, ctx.eps,
                                         _dt(k_t), None, _stream()), "slot_attn_bwd")
        if acc is not None:
            acc.pending -= 1
            if acc.pending > 0:
                return None, None, dq, None, None                 # the sum leaves with the last iteration's node
            acc.dk = acc.dv = None
            acc.total = 0
        return dk, dv, dq, None, None


def slot_attn_step(k_t, v_t, q, eps, acc=None):
    """-> (updates [B,K,D], attn_vis [B,N,K]).  acc: a SlotKVGrad shared by the iterations of one frame (optional)."""
    return _SlotAttnFn.apply(k_t, v_t, q, eps, acc)


class _GruGatesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, gh, h):
        _need_gpu(gi, gh, h)
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        R, D = h.shape
        hn = torch.empty_like(h)
        _lib.check(_lib.lib().focus_gru_gates_fwd(_p(gi), _p(gh), _p(h), _p(hn), None, None, R, D, _dt(h), _stream()),
                   "gru_fwd")
        ctx.save_for_backward(gi, gh, h)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        gi, gh, h = ctx.saved_tensors
        R, D = h.shape
        dhn = dhn.contiguous()
        dgi, dgh, dh = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        _lib.check(_lib.lib().focus_gru_gates_bwd(_p(gi), _p(gh), _p(h), _p(dhn), _p(dgi), _p(dgh), _p(dh), None, R, D,
                                                  _dt(h), _stream()), "gru_bwd")
        return dgi, dgh, dh


def _stacked_cat(ws, dtype, transposed):
    """The bf16 shadows of several [C, Cin] weights concatenated into one NT-GEMM B operand: rows stacked [sum C, Cin]
    (forward), or the transposes side by side [Cin, sum C] (d(input)); cached like the shadows themselves."""
    def build(out):
        parts = [shadow(w, dtype, transposed=transposed) for w in ws]
        dim = 1 if transposed else 0
        return torch.cat(parts, dim=dim, out=out) if out is not None else torch.cat(parts, dim=dim).contiguous()
    return _stacked_get(tuple(id(w) for w in ws) + (dtype, transposed, "cat"), tuple(ws), None, build)


class _LinearQKVFn(torch.autograd.Function):
    """Three bias-free Linears of one input (transformer.py:33-35, self-attention) as ONE product: the outputs are the
    column blocks of a [rows, 3C] matrix (strided views; ops.small_attention reads them in place and returns its three
    gradients as the blocks of one matrix again), so d(input) = [dq|dk|dv] . [Wq;Wk;Wv] is one product too."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        _need_gpu(x, wq)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        C = wq.shape[0]
        y = mm_nt(x2, _stacked_cat((wq, wk, wv), x.dtype, False)).view(*shp[:-1], 3 * C)
        ctx.save_for_backward(x2, wq, wk, wv)
        ctx.shp = shp
        ds = [_defer_open("linear", (w, None), x2.shape[0], ctx.needs_input_grad[1 + i]) for i, w in enumerate((wq, wk, wv))]
        if any(d is None for d in ds):
            for d in ds:
                if d is not None:
                    d.pending -= 1
            ds = None
        ctx.defer = ds
        return y[..., :C], y[..., C:2 * C], y[..., 2 * C:]

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x2, wq, wk, wv = ctx.saved_tensors
        C = wq.shape[0]
        ds = [t.reshape(-1, C) for t in (dq, dk, dv)]
        es = ds[0].element_size()
        joint = (all(t.stride() == (3 * C, 1) for t in ds) and ds[1].data_ptr() == ds[0].data_ptr() + C * es
                 and ds[2].data_ptr() == ds[1].data_ptr() + C * es and ds[0].dtype == torch.bfloat16 and C % 64 == 0)
        if joint:
            dy = torch.as_strided(ds[0], (ds[0].shape[0], 3 * C), (3 * C, 1))
            dx = mm_nt(dy, _stacked_cat((wq, wk, wv), dy.dtype, True))
        else:
            ds = [t if t.is_contiguous() else t.contiguous() for t in ds]
            dx = _dx_from(ds[0], wq, ds[0].dtype)
            for t, w in ((ds[1], wk), (ds[2], wv)):
                dx = dx + _dx_from(t, w, t.dtype)
        dws = [None, None, None]
        for i, (t, w) in enumerate(zip(ds, (wq, wk, wv))):
            if ctx.defer:
                done = _defer_close(ctx.defer[i], (t, x2))
                if done is not None:
                    dws[i] = done[0]
            else:
                dws[i] = linear_wgrad(t if t.stride(0) % 8 == 0 else t.contiguous(), x2, False)[0]
        return dx.reshape(ctx.shp), dws[0], dws[1], dws[2]


def linear_qkv(x, wq, wk, wv):
    """(x.wq^T, x.wk^T, x.wv^T) as the column blocks of one product (strided views of a [..., 3C] tensor)."""
    return _LinearQKVFn.apply(x, wq, wk, wv)


def _stacked_pair(wa, wb, dtype, transposed):
    """The bf16 shadows of two equally shaped weights as one [2, ...] tensor (the B operands of a batch-2 product);
    cached like the shadows themselves (version, storage, shadow generation)."""
    def build(out):
        parts = [shadow(wa, dtype, transposed=transposed), shadow(wb, dtype, transposed=transposed)]
        return torch.stack(parts, 0, out=out) if out is not None else torch.stack(parts, 0).contiguous()
    return _stacked_get((id(wa), id(wb), dtype, transposed), (wa, wb), None, build)


class _GruCellFn(torch.autograd.Function):
    """nn.GRUCell (STEVE/utils.py:107-118) with its two Linear products as ONE batch-2 launch each way: forward
    [gi | gh] = [x | h] . [W_ih | W_hh]^T (the operands of the two halves are separate tensors: the batch stride of A is
    simply their distance), backward [dx | dh] = [dgi | dgh] . [W_ih | W_hh] + [0 | dhn z].  The gate kernel adds the
    biases.  Per application: 2 launches forward, 2 backward (5 and 7 before, with autograd's accumulation add)."""

    @staticmethod
    def forward(ctx, x, h, w_ih, w_hh, b_ih, b_hh):
        R, D = h.shape
        G = 3 * D
        g = torch.empty(2, R, G, device=x.device, dtype=x.dtype)
        gemm(R, G, D, (x, 0), (D, 1, 0, (h.data_ptr() - x.data_ptr()) // x.element_size()),
             (_stacked_pair(w_ih, w_hh, x.dtype, False), 0), (1, D, 0, G * D), (g, 0), (G, 1, 0, R * G), batch=(1, 2))
        hn = torch.empty_like(h)
        _lib.check(_lib.lib().focus_gru_gates_fwd(_p(g[0]), _p(g[1]), _p(h), _p(hn), _p(b_ih), _p(b_hh), R, D, _dt(h),
                                                  _stream()), "gru_fwd")
        ctx.save_for_backward(x, h, g, w_ih, w_hh, b_ih, b_hh)
        di = _defer_open("linear", (w_ih, b_ih), R, ctx.needs_input_grad[2])
        dh = _defer_open("linear", (w_hh, b_hh), R, ctx.needs_input_grad[3]) if di is not None else None
        if di is not None and dh is None:
            di.pending -= 1
            di = None
        ctx.defer = (di, dh) if di is not None else None
        return hn

    @staticmethod
    def backward(ctx, dhn):
        x, h, g, w_ih, w_hh, b_ih, b_hh = ctx.saved_tensors
        R, D = h.shape
        G = 3 * D
        dhn = dhn.contiguous()
        dg = torch.empty_like(g)
        res = torch.empty(2, R, D, device=h.device, dtype=h.dtype)
        _lib.check(_lib.lib().focus_gru_gates_bwd(_p(g[0]), _p(g[1]), _p(h), _p(dhn), _p(dg[0]), _p(dg[1]), _p(res[1]),
                                                  _p(res[0]), R, D, _dt(h), _stream()), "gru_bwd")
        dxh = torch.empty_like(res)
        gemm(R, D, G, (dg, 0), (G, 1, 0, R * G), (_stacked_pair(w_ih, w_hh, h.dtype, True), 0), (1, G, 0, D * G),
             (dxh, 0), (D, 1, 0, R * D), batch=(1, 2), residual=(res, 0))
        dws = [None, None, None, None]
        if ctx.defer:
            done = _defer_close(ctx.defer[0], (dg[0], x))
            if done is not None:
                dws[0], dws[2] = done
            done = _defer_close(ctx.defer[1], (dg[1], h))
            if done is not None:
                dws[1], dws[3] = done
        else:
            dws[0], dws[2] = linear_wgrad(dg[0], x, True)
            dws[1], dws[3] = linear_wgrad(dg[1], h, True)
        return dxh[0], dxh[1], dws[0], dws[1], dws[2], dws[3]


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """nn.GRUCell.  x, h [R,D].  bf16 with equal input and hidden widths (the slot update): _GruCellFn; otherwise two
    Linear GEMMs + the gate kernel."""
    if (x.dtype == torch.bfloat16 and x.shape == h.shape and x.is_contiguous() and h.is_contiguous() and x.shape[1] % 32 == 0
            and b_ih is not None and b_hh is not None and (h.data_ptr() - x.data_ptr()) % 16 == 0 and x.is_cuda):
        return _GruCellFn.apply(x, h, w_ih, w_hh, b_ih, b_hh)
    return _GruGatesFn.apply(linear(x, w_ih, b_ih), linear(h, w_hh, b_hh), h)


# --------------------------------------------------------------------------------------------------
# The per-iteration tail of the slot update as one forward launch (csrc/slot_tail.hip):
#   [gru] hn = GRUCell(upd, h)   [mlp] s = hn + W2 relu(W1 LN1(hn) + b1) + b2   [q] q = Wq LN2(slots)
# --------------------------------------------------------------------------------------------------
class SlotTailParams:
    """The parameters of SlotAttentionVideo's recurrent tail (steve.py:35-49), in the order _SlotTailFn takes them."""

    def __init__(self, gru, norm_mlp, mlp, norm_slots, project_q):
        self.tensors = (gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh, norm_mlp.weight, norm_mlp.bias,
                        mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, norm_slots.weight, norm_slots.bias,
                        project_q.weight)
        self.eps = (norm_mlp.eps, norm_slots.eps)


def slot_tail_ok(x, params):
    """The fused tail takes bf16 rows of width 192 with a 768-wide MLP (the BASELINE shape); FOCUS_SLOT_TAIL=0 disables it."""
    if _os.environ.get("FOCUS_SLOT_TAIL", "1") == "0" or x.dtype != torch.bfloat16 or not x.is_cuda:
        return False
    D, H = params.tensors[0].shape[1], params.tensors[6].shape[0]
    return bool(_lib.lib().focus_slot_tail_ok(D, H, BF16)) and all(t is not None for t in params.tensors)


def _tail_wgrad(stash, w, has_b, dy, x):
    """(dw, db) of one Linear application inside the tail: deferred (stacked over the applications) when a stash is open."""
    if stash is not None:
        done = _defer_close(stash, (dy, x))
        return done if done is not None else (None, None)
    return linear_wgrad(dy, x, bool(has_b))


class _SlotTailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, upd, h, eps1, eps2, do_gru, do_mlp, do_q, w_ih, w_hh, b_ih, b_hh, g1, be1, w1, b1, w2, b2, g2, be2, wq):
        _need_gpu(h, wq)
        h = h.contiguous()
        R, D = h.shape
        H = w1.shape[0]
        dev, dt = h.device, h.dtype
        a = _lib.SlotTailArgs()
        a.R, a.D, a.H, a.do_gru, a.do_mlp, a.do_q = R, D, H, int(do_gru), int(do_mlp), int(do_q)
        a.ln1_eps, a.ln2_eps = eps1, eps2
        keep = [h]
        new = lambda *s, dtype=dt: torch.empty(*s, device=dev, dtype=dtype)
        a.h = h.data_ptr()
        sv = {}
        if do_gru:
            upd = upd.contiguous()
            keep.append(upd)
            a.upd = upd.data_ptr()
            ws = [shadow(w_ih, dt), shadow(w_hh, dt)]
            keep += ws
            a.w_ih, a.w_hh, a.b_ih, a.b_hh = ws[0].data_ptr(), ws[1].data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
            sv["g"], sv["hn"] = new(2, R, 3 * D), new(R, D)
            a.g, a.hn = sv["g"].data_ptr(), sv["hn"].data_ptr()
        if do_mlp:
            ws = [shadow(w1, dt), shadow(w2, dt)]
            keep += ws
            a.ln1_g, a.ln1_b, a.w1, a.b1, a.w2, a.b2 = (g1.data_ptr(), be1.data_ptr(), ws[0].data_ptr(), b1.data_ptr(),
                                                        ws[1].data_ptr(), b2.data_ptr())
            sv["y"], sv["a"], sv["s"] = new(R, D), new(R, H), new(R, D)
            sv["mean1"], sv["rstd1"] = new(R, dtype=torch.float32), new(R, dtype=torch.float32)
            a.y, a.a, a.s, a.mean1, a.rstd1 = (sv["y"].data_ptr(), sv["a"].data_ptr(), sv["s"].data_ptr(), sv["mean1"].data_ptr(),
                                               sv["rstd1"].data_ptr())
        if do_q:
            wqs = shadow(wq, dt)
            keep.append(wqs)
            a.ln2_g, a.ln2_b, a.wq = g2.data_ptr(), be2.data_ptr(), wqs.data_ptr()
            sv["sn"], sv["q"] = new(R, D), new(R, D)
            sv["mean2"], sv["rstd2"] = new(R, dtype=torch.float32), new(R, dtype=torch.float32)
            a.sn, a.q, a.mean2, a.rstd2 = sv["sn"].data_ptr(), sv["q"].data_ptr(), sv["mean2"].data_ptr(), sv["rstd2"].data_ptr()
        _lib.check(_lib.lib().focus_slot_tail_fwd(ctypes.byref(a), _stream()), "slot_tail_fwd")
        out = sv["s"] if do_mlp else (sv["hn"] if do_gru else h.view_as(h))
        ctx.flags = (do_gru, do_mlp, do_q)
        ctx.names = [k for k in ("g", "hn", "y", "a", "s", "mean1", "rstd1", "sn", "mean2", "rstd2") if k in sv]
        ctx.save_for_backward(upd if do_gru else None, h, w_ih, w_hh, g1, w1, w2, g2, wq, *[sv[k] for k in ctx.names])
        ctx.has_b = (b_ih is not None, b1 is not None, b2 is not None)
        ni = ctx.needs_input_grad
        # parameter gradients: stacked over all applications of the slot loop when ops.deferred_wgrads is open
        ctx.st = {}
        if do_gru:
            ctx.st["ih"] = _defer_open("linear", (w_ih, b_ih), R, ni[7])
            ctx.st["hh"] = _defer_open("linear", (w_hh, b_hh), R, ni[8])
        if do_mlp:
            ctx.st["ln1"] = _defer_open("ln", (g1, be1), R, ni[11])
            ctx.st["w1"] = _defer_open("linear", (w1, b1), R, ni[13])
            ctx.st["w2"] = _defer_open("linear", (w2, b2), R, ni[15])
        if do_q:
            ctx.st["ln2"] = _defer_open("ln", (g2, be2), R, ni[17])
            ctx.st["wq"] = _defer_open("linear", (wq, None), R, ni[19])
        return out, (sv["q"] if do_q else None)

    @staticmethod
    def backward(ctx, dout, dq):
        do_gru, do_mlp, do_q = ctx.flags
        saved = ctx.saved_tensors
        upd, h, w_ih, w_hh, g1, w1, w2, g2, wq = saved[:9]
        sv = dict(zip(ctx.names, saved[9:]))
        R, D = h.shape
        dt = h.dtype
        L = _lib.lib()
        grads = {}

        def ln_bwd(dy, x, gamma, mean, rstd, dres, key):
            nblk = L.focus_layernorm_bwd_blocks(R)
            partial = torch.empty(2, nblk, D, device=h.device, dtype=torch.float32)
            dx = torch.empty_like(x)
            st = ctx.st.get(key)
            dg = db = None
            if st is None:
                dg, db = torch.empty(D, device=h.device, dtype=torch.float32), torch.empty(D, device=h.device, dtype=torch.float32)
            _lib.check(L.focus_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres) if dres is not None else None,
                                             _p(dx), _p(dg) if dg is not None else None, _p(db) if db is not None else None,
                                             _p(partial), R, D, _dt(x), _stream()), "layernorm_bwd")
            if st is not None:
                done = _defer_close(st, partial)
                dg, db = done if done is not None else (None, None)
            return dx, dg, db

        if _SLOT_TAIL_BWD and dt == torch.bfloat16 and bool(L.focus_slot_tail_ok(D, w1.shape[0], _dt(h))) \
                and (do_gru or (do_q and dq is not None)):
            return _SlotTailFn._backward_fused(ctx, dout, dq, upd, h, w_ih, w_hh, g1, w1, w2, g2, wq, sv)
        dcur = dout.contiguous() if dout is not None else None
        if do_q and dq is not None:
            dq = dq.contiguous()
            cur = sv["s"] if do_mlp else (sv["hn"] if do_gru else h)
            dsn = _dx_from(dq, wq, dt)
            grads["wq"] = _tail_wgrad(ctx.st.get("wq"), wq, None, dq, sv["sn"])[0]
            dcur, grads["g2"], grads["be2"] = ln_bwd(dsn, cur, g2, sv["mean2"], sv["rstd2"], dcur, "ln2")
        if dcur is None:
            dcur = torch.zeros_like(h)
        if do_mlp:
            ds = dcur
            grads["w2"], grads["b2"] = _tail_wgrad(ctx.st.get("w2"), w2, ctx.has_b[2], ds, sv["a"])
            dz = _dx_from(ds, w2, dt, aux=sv["a"], epilogue=_DEPI[EPI_RELU])
            grads["w1"], grads["b1"] = _tail_wgrad(ctx.st.get("w1"), w1, ctx.has_b[1], dz, sv["y"])
            dy1 = _dx_from(dz, w1, dt)
            dhn, grads["g1"], grads["be1"] = ln_bwd(dy1, sv["hn"], g1, sv["mean1"], sv["rstd1"], ds, "ln1")
        else:
            dhn = dcur
        dupd = None
        if do_gru:
            g = sv["g"]
            G = 3 * D
            dg = torch.empty_like(g)
            res = torch.empty(2, R, D, device=h.device, dtype=dt)
            _lib.check(L.focus_gru_gates_bwd(_p(g[0]), _p(g[1]), _p(h), _p(dhn), _p(dg[0]), _p(dg[1]), _p(res[1]), _p(res[0]), R, D,
                                             _dt(h), _stream()), "gru_bwd")
            dxh = torch.empty_like(res)
            gemm(R, D, G, (dg, 0), (G, 1, 0, R * G), (_stacked_pair(w_ih, w_hh, dt, True), 0), (1, G, 0, D * G),
                 (dxh, 0), (D, 1, 0, R * D), batch=(1, 2), residual=(res, 0))
            grads["w_ih"], grads["b_ih"] = _tail_wgrad(ctx.st.get("ih"), w_ih, ctx.has_b[0], dg[0], upd)
            grads["w_hh"], grads["b_hh"] = _tail_wgrad(ctx.st.get("hh"), w_hh, ctx.has_b[0], dg[1], h)
            dupd, dh = dxh[0], dxh[1]
        else:
            dh = dhn
        gg = grads.get
        return (dupd, dh, None, None, None, None, None, gg("w_ih"), gg("w_hh"), gg("b_ih"), gg("b_hh"), gg("g1"), gg("be1"), gg("w1"),
                gg("b1"), gg("w2"), gg("b2"), gg("g2"), gg("be2"), gg("wq"))


def _slot_tail_backward_fused(ctx, dout, dq, upd, h, w_ih, w_hh, g1, w1, w2, g2, wq, sv):
    """The backward of the tail as ONE launch (focus_slot_tail_bwd): dX chain in the kernel, the dY rows of every weight gradient
    written out and handed to the (deferred) weight-gradient machinery exactly as the composed backward does."""
    do_gru, do_mlp, do_q = ctx.flags
    do_q = do_q and dq is not None
    R, D = h.shape
    H = w1.shape[0]
    dev, dt = h.device, h.dtype
    L = _lib.lib()
    new = lambda *s_, dtype=dt: torch.empty(*s_, device=dev, dtype=dtype)
    a = _lib.SlotTailBwdArgs()
    a.R, a.D, a.H, a.do_gru, a.do_mlp, a.do_q = R, D, H, int(do_gru), int(do_mlp), int(do_q)
    keep = []
    if dout is not None:
        dout = dout.contiguous()
        a.dout = dout.data_ptr()
    nblk = L.focus_slot_tail_bwd_blocks(R)
    ds = dz = dg = dupd = part1 = part2 = None
    dh = new(R, D)
    a.dh, a.h = dh.data_ptr(), h.data_ptr()
    if do_q:
        dq = dq.contiguous()
        cur = sv["s"] if do_mlp else (sv["hn"] if do_gru else h)
        wqt = shadow(wq, dt, transposed=True)
        part2 = new(2, nblk, D, dtype=torch.float32)
        keep += [wqt, cur]
        a.dq, a.cur, a.mean2, a.rstd2, a.ln2_g, a.wq_t, a.part2 = (dq.data_ptr(), cur.data_ptr(), sv["mean2"].data_ptr(),
                                                                   sv["rstd2"].data_ptr(), g2.data_ptr(), wqt.data_ptr(),
                                                                   part2.data_ptr())
    if do_q or do_mlp:
        ds = new(R, D)
        a.ds = ds.data_ptr()
    if do_mlp:
        w1t, w2t = shadow(w1, dt, transposed=True), shadow(w2, dt, transposed=True)
        dz, part1 = new(R, H), new(2, nblk, D, dtype=torch.float32)
        keep += [w1t, w2t]
        a.a, a.hn, a.mean1, a.rstd1, a.ln1_g, a.w1_t, a.w2_t, a.dz, a.part1 = (
            sv["a"].data_ptr(), sv["hn"].data_ptr(), sv["mean1"].data_ptr(), sv["rstd1"].data_ptr(), g1.data_ptr(), w1t.data_ptr(),
            w2t.data_ptr(), dz.data_ptr(), part1.data_ptr())
    if do_gru:
        wit, wht = shadow(w_ih, dt, transposed=True), shadow(w_hh, dt, transposed=True)
        dg, dupd = new(2, R, 3 * D), new(R, D)
        keep += [wit, wht]
        a.g, a.w_ih_t, a.w_hh_t, a.dg, a.dupd = sv["g"].data_ptr(), wit.data_ptr(), wht.data_ptr(), dg.data_ptr(), dupd.data_ptr()
    # [R, D] scratch between the launches of the staged form: dsn | dy1 | z * dhn
    ws = new(3, R, D)
    keep.append(ws)
    a.ws_dsn, a.ws_dy1, a.ws_res = ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr()
    _lib.check(L.focus_slot_tail_bwd(ctypes.byref(a), _stream()), "slot_tail_bwd")
    grads = {}

    def ln_done(key, part):
        st = ctx.st.get(key)
        if st is not None:
            done = _defer_close(st, part)
            return done if done is not None else (None, None)
        both = part.sum(1)
        return both[0], both[1]

    if do_q:
        grads["wq"] = _tail_wgrad(ctx.st.get("wq"), wq, None, dq, sv["sn"])[0]
        grads["g2"], grads["be2"] = ln_done("ln2", part2)
    if do_mlp:
        grads["w2"], grads["b2"] = _tail_wgrad(ctx.st.get("w2"), w2, ctx.has_b[2], ds, sv["a"])
        grads["w1"], grads["b1"] = _tail_wgrad(ctx.st.get("w1"), w1, ctx.has_b[1], dz, sv["y"])
        grads["g1"], grads["be1"] = ln_done("ln1", part1)
    if do_gru:
        grads["w_ih"], grads["b_ih"] = _tail_wgrad(ctx.st.get("ih"), w_ih, ctx.has_b[0], dg[0], upd)
        grads["w_hh"], grads["b_hh"] = _tail_wgrad(ctx.st.get("hh"), w_hh, ctx.has_b[0], dg[1], h)
    gg = grads.get
    return (dupd, dh, None, None, None, None, None, gg("w_ih"), gg("w_hh"), gg("b_ih"), gg("b_hh"), gg("g1"), gg("be1"), gg("w1"),
            gg("b1"), gg("w2"), gg("b2"), gg("g2"), gg("be2"), gg("wq"))


_SlotTailFn._backward_fused = staticmethod(_slot_tail_backward_fused)
# Measured: neither form of focus_slot_tail_bwd beats the five small launches of the composed backward under
# graph replay -- one workgroup per 16 rows streams 1.1 MB of weights by itself (35 us per call; whole step 17.4 -> 18.7 ms),
# the five right-sized launches (FOCUS_SLOT_TAIL_STAGED=1, the default form) redo the LayerNorm backward per column tile and
# wait on scalar gate loads (16.0 -> 17.0 ms).  Both win only where launches are the cost (eager: 38 -> 35 ms).  Off by default.
_SLOT_TAIL_BWD = _os.environ.get("FOCUS_SLOT_TAIL_BWD", "0") != "0"


def slot_tail(upd, h, params, gru=True, mlp=True, q=True):
    """-> (slots [R,D], q [R,D] or None).  upd, h [R,D] bf16 (upd is ignored without gru); see the section comment."""
    return _SlotTailFn.apply(upd, h, float(params.eps[0]), float(params.eps[1]), bool(gru), bool(mlp), bool(q), *params.tensors)
