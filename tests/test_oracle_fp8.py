"""oracle/fp8.py (OCP e4m3fn restatement used to check the fp8-weight GEMM path) against the format's known answers and
against torch's own float8_e4m3fn conversion on CPU (in range: torch does not saturate)."""
import numpy as np
import torch

from oracle import fp8


def test_decode_known_answers():
    t = fp8.decode_table()
    assert t[0x00] == 0.0 and t[0x01] == 2.0 ** -9 and t[0x07] == 7 * 2.0 ** -9          # subnormals
    assert t[0x08] == 2.0 ** -6 and t[0x38] == 1.0 and t[0x3C] == 1.5 and t[0x40] == 2.0
    assert t[0x7E] == 448.0 and np.isnan(t[0x7F]) and t[0xFE] == -448.0 and np.isnan(t[0xFF])
    assert np.all(np.diff(t[:127]) > 0)


def test_encode_round_trip_ties_and_saturation():
    t = fp8.decode_table()
    codes = np.array([c for c in range(256) if c not in (0x7F, 0xFF)], dtype=np.uint8)
    assert np.array_equal(fp8.encode(t[codes]) & 0x7F, codes & 0x7F)                    # every value maps to itself
    assert fp8.encode(np.float32(1.0625))[()] == 0x38        # tie between 1.0 (0x38, even) and 1.125 (0x39): even
    assert fp8.encode(np.float32(1.1875))[()] == 0x3A        # tie between 1.125 (0x39) and 1.25 (0x3A): even
    assert fp8.encode(np.float32(1e9))[()] == 0x7E and fp8.encode(np.float32(-1e9))[()] == 0xFE
    assert fp8.encode(np.float32(2.0 ** -10))[()] == 0x00 and fp8.encode(np.float32(3 * 2.0 ** -10))[()] == 0x02   # ties at the bottom


def test_encode_matches_torch_float8_in_range():
    g = np.random.default_rng(0)
    x = np.concatenate([g.standard_normal(200000).astype(np.float32) * s for s in (1e-3, 0.05, 1.0, 100.0)])
    x = x[np.abs(x) <= 448.0]
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = fp8.encode(x)
    z = (ref & 0x7F) == 0                                     # +-0: sign handling of zeros may differ, value does not
    assert np.array_equal(got[~z], ref[~z]) and np.all((got[z] & 0x7F) == 0)


def test_per_tensor_quantisation_error_bound():
    g = np.random.default_rng(1)
    w = (g.standard_normal((384, 768)) * 0.02).astype(np.float32)
    q = fp8.fake_quant(w)
    amax = np.abs(w).max()
    # 3 mantissa bits: relative error <= 2^-4 for normal values, absolute <= scale * 2^-10 below the normal range
    err = np.abs(q - w)
    assert np.all(err <= np.maximum(np.abs(w) * 2.0 ** -4, amax / 448.0 * 2.0 ** -10) * 1.0001)
    assert np.abs(q).max() == amax or abs(np.abs(q).max() - amax) <= amax * 2 ** -4
