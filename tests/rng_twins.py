"""numpy twins of the counter-based draws made inside csrc/gumbel.hip (build-owned: the reference draws with torch's
generator, whose stream no other implementation reproduces; parity runs pass the reference's draws in instead).
The GPU tests hold the kernels to these functions; tests/test_rng_twins.py checks the functions' statistics on the CPU."""
import numpy as np

_M32 = np.uint64(0xffffffff)


def mix32(x):
    """lowbias32: the 32-bit finaliser both generators are built on (mix32 in csrc/gumbel.hip)."""
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    return x


def exp1_draws(s0, s1, R, V):
    """exp1_draw / gumbel_row: E[r, c] ~ Exp(1) from 23 hashed bits, u = (k + 1/2) / 2^23, E = -ln 2 * log2(u)."""
    rows = np.arange(R, dtype=np.uint64)
    row_key = (mix32((rows & _M32) ^ np.uint64(s0)) + (rows >> np.uint64(32))) & _M32
    cols = (np.arange(V, dtype=np.uint64) * np.uint64(0x9E3779B9)) & _M32
    h = mix32(row_key[:, None] ^ cols[None, :] ^ np.uint64(s1))
    u = ((h >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    return (-np.float32(0.6931471805599453) * np.log2(u)).astype(np.float32)


def keep_mask(s0, s1, n, thr):
    """dropout_add_kernel: element i survives when its 16 hashed bits are >= thr; one 32-bit hash per element pair."""
    pair = np.arange(n // 2, dtype=np.uint64)
    h = mix32((pair & _M32) ^ np.uint64(s0)) ^ (((pair >> np.uint64(32)) * np.uint64(0x9E3779B9) + np.uint64(s1)) & _M32)
    lo, hi = h & np.uint64(0xffff), h >> np.uint64(16)
    return np.stack([lo >= thr, hi >= thr], axis=1).reshape(-1)
