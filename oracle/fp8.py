"""OCP FP8 E4M3 (`e4m3fn`: 1 sign, 4 exponent bits with bias 7, 3 mantissa bits, no infinities, 0x7F / 0xFF = NaN, largest
finite value 448) in numpy -- TEST INFRASTRUCTURE ONLY (the checker for the fp8-weight GEMM path of focus_amd; the product
never imports this).  The format is the published OCP 8-bit floating point specification (OFP8 rev 1.0, table 1); gfx950
implements this encoding (not MI300's e4m3fnuz).  Per-tensor scaling as used by the product (csrc/fp8.hip):
    inv = 448 / amax  (fp32),   code = RNE_e4m3(w * inv)  (fp32 product),   w_deq = decode(code) * (amax / 448)
"""
import numpy as np

E4M3_MAX = 448.0


def decode_table():
    """float32 value of each of the 256 codes (NaN for 0x7F and 0xFF)."""
    t = np.empty(256, dtype=np.float32)
    for c in range(256):
        s = -1.0 if c & 0x80 else 1.0
        e, m = (c >> 3) & 0xF, c & 7
        if e == 0xF and m == 7:
            v = np.nan
        elif e == 0:
            v = s * m * 2.0 ** -9                      # subnormal: m/8 * 2^(1-7)
        else:
            v = s * (1.0 + m / 8.0) * 2.0 ** (e - 7)
        t[c] = v
    return t


_TABLE = decode_table()
_POS = _TABLE[:127].astype(np.float64)                 # codes 0x00..0x7E ascending: 0 .. 448


def encode(x):
    """Round-to-nearest-even into e4m3fn codes (uint8), saturating at +-448; NaN -> 0x7F | sign."""
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x).astype(np.float64)
    a = np.minimum(a, E4M3_MAX)
    hi = np.searchsorted(_POS, a, side="left")          # first code with value >= a
    hi = np.clip(hi, 0, 126)
    lo = np.clip(hi - 1, 0, 126)
    dlo, dhi = a - _POS[lo], _POS[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((hi & 1) == 0))      # ties to the even code (even mantissa)
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    code = np.where(np.isnan(x), 0x7F, code).astype(np.uint8)
    sign = (np.signbit(x).astype(np.uint8) << 7)
    return (code | sign).astype(np.uint8)


def decode(code):
    return _TABLE[np.asarray(code, dtype=np.uint8)]


def quantize_per_tensor(w):
    """-> (codes uint8, scale float32) with w ~= decode(codes) * scale; scale = amax / 448 (1.0 for an all-zero tensor)."""
    w = np.asarray(w, dtype=np.float32)
    amax = np.float32(np.abs(w).max()) if w.size else np.float32(0)
    if not np.isfinite(amax) or amax == 0:
        return encode(np.zeros_like(w)), np.float32(1.0)
    inv = np.float32(E4M3_MAX) / amax                   # fp32 division, as on the device
    scale = amax / np.float32(E4M3_MAX)
    return encode(w * inv), np.float32(scale)


def fake_quant(w):
    """w rounded through per-tensor-scaled e4m3 and back (float32): the weights the fp8 GEMM actually multiplies by."""
    codes, scale = quantize_per_tensor(w)
    return decode(codes) * scale
