"""ctypes binding of libfocus_amd.so.  argtypes are derived from include/focus_amd.h itself, so the
Python side cannot drift from the C ABI.  Loading fails loudly: there is no CPU or PyTorch fallback."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(_HERE, "..", "include", "focus_amd.h")
DEBUG_HEADER = os.path.join(_HERE, "csrc", "focus_debug.h")     # test-suite probes, not part of the operator ABI
LIB_PATH = os.path.join(_HERE, "lib", "libfocus_amd.so")

F32, BF16, FP8_E4M3 = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_RELU, EPI_TANH, EPI_DGELU, EPI_DRELU, EPI_DTANH = range(7)


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("batch0", ctypes.c_int32), ("batch1", ctypes.c_int32),
        ("A", ctypes.c_void_p), ("rsA", ctypes.c_int64), ("csA", ctypes.c_int64),
        ("bsA0", ctypes.c_int64), ("bsA1", ctypes.c_int64),
        ("B", ctypes.c_void_p), ("rsB", ctypes.c_int64), ("csB", ctypes.c_int64),
        ("bsB0", ctypes.c_int64), ("bsB1", ctypes.c_int64),
        ("C", ctypes.c_void_p), ("rsC", ctypes.c_int64), ("csC", ctypes.c_int64),
        ("bsC0", ctypes.c_int64), ("bsC1", ctypes.c_int64),
        ("bias", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("aux", ctypes.c_void_p),
        ("alpha", ctypes.c_float), ("accumulate", ctypes.c_int32), ("epilogue", ctypes.c_int32),
        ("dtype_ab", ctypes.c_int32), ("dtype_c", ctypes.c_int32),
        ("dtype_b", ctypes.c_int32), ("pad_", ctypes.c_int32), ("b_scale", ctypes.c_void_p),
    ]


class WgradItem(ctypes.Structure):
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("ld_dy", ctypes.c_int64), ("ld_x", ctypes.c_int64),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class SlotTailArgs(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("R", "D", "H", "do_gru", "do_mlp", "do_q")] +
                [("ln1_eps", ctypes.c_float), ("ln2_eps", ctypes.c_float)] +
                [(n, ctypes.c_void_p) for n in ("upd", "h", "w_ih", "w_hh", "b_ih", "b_hh", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2",
                                                "ln2_g", "ln2_b", "wq", "g", "hn", "y", "mean1", "rstd1", "a", "s", "sn", "mean2",
                                                "rstd2", "q")])


class SlotTailBwdArgs(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("R", "D", "H", "do_gru", "do_mlp", "do_q")] +
                [(n, ctypes.c_void_p) for n in ("dout", "dq", "h", "g", "hn", "a", "cur", "mean1", "rstd1", "mean2", "rstd2",
                                                "ln1_g", "ln2_g", "w_ih_t", "w_hh_t", "w1_t", "w2_t", "wq_t", "dupd", "dh", "ds",
                                                "dz", "dg", "part1", "part2", "ws_dsn", "ws_dy1", "ws_res")])


class FlashArgs(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("q", "k", "v", "out", "lse", "dout", "delta", "dq", "dk", "dv", "seed")] +
                [(n, ctypes.c_int64) for n in ("ldq", "ldk", "ldv", "ldo", "lddo", "lddq", "lddk", "lddv",
                                               "bsq", "bsk", "bsv", "bso", "bsdo", "bsdq", "bsdk", "bsdv")] +
                [(n, ctypes.c_int32) for n in ("B", "heads", "Nq", "Nk", "d", "dtype", "causal")] +
                [("drop_thr", ctypes.c_uint32), ("scale", ctypes.c_float), ("pad_", ctypes.c_int32)])


_CTYPE = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
          "size_t": ctypes.c_size_t, "float": ctypes.c_float, "double": ctypes.c_double}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the public header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", src, flags=re.S)
    src = re.sub(r"enum \w+ \{.*?\};", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(focus_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "char" in ret:
            restype = ctypes.c_char_p
        elif "size_t" in ret:
            restype = ctypes.c_size_t
        else:
            restype = ctypes.c_int
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.POINTER(GemmDesc) if "focus_gemm_desc" in a else ctypes.c_void_p)
                else:
                    ty = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPE[ty])
        out[name] = (restype, argtypes)
    return out


_lib = None


def lib():
    """The loaded library with typed entry points (raises if the HIP extension has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "focus_amd: %s is missing -- build it with `python -m focus_amd.build` "
                "(there is no CPU/PyTorch fallback for the hot path)" % LIB_PATH)
        # torch first: libfocus_amd.so needs libamdhip64, and the process must end up with ONE HIP runtime -- the copy
        # PyTorch-ROCm ships and initialises.  Loaded the other way round (this library before torch) the kernels
        # register with /opt/rocm's runtime while the streams come from torch's: every launch then fails.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        decls = dict(parse_header())
        decls.update(parse_header(DEBUG_HEADER))
        for name, (restype, argtypes) in decls.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        if L.focus_abi_version() != 2:
            raise RuntimeError("focus_amd: ABI version mismatch")
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        raise RuntimeError("focus_amd %s failed: %s (%d)" % (what, lib().focus_strerror(status).decode(), status))
