"""Statistics of the counter-based draws the Gumbel and dropout kernels make (tests/rng_twins.py = their numpy twins, held
to the kernels bit for bit by tests/test_gpu_gumbel.py): the right distribution, no visible structure along rows, columns or
between the two streams of a seed."""
import numpy as np

from rng_twins import exp1_draws, keep_mask


def test_exp1_draws_are_exponential():
    e = exp1_draws(123456789, 987654321, 512, 4096).astype(np.float64)
    n = e.size
    assert e.min() > 0.0 and e.max() < 17.0                       # u in [2^-24, 1 - 2^-24]
    assert abs(e.mean() - 1.0) < 5.0 / np.sqrt(n)                 # Exp(1): mean 1, variance 1
    assert abs(e.var() - 1.0) < 5.0 * np.sqrt(8.0 / n)            # var of the sample variance of Exp(1) = 8 / n
    # quantiles of Exp(1): P(E > t) = exp(-t)
    for t in (0.1, 0.5, 1.0, 2.0, 4.0):
        pt = np.exp(-t)
        assert abs((e > t).mean() - pt) < 5.0 * np.sqrt(pt * (1 - pt) / n)
    # no correlation between neighbours along a row or a column
    c = e - 1.0
    assert abs((c[:, 1:] * c[:, :-1]).mean()) < 5.0 / np.sqrt(n)
    assert abs((c[1:] * c[:-1]).mean()) < 5.0 / np.sqrt(n)


def test_exp1_streams_differ_by_seed_and_repeat_by_seed():
    a = exp1_draws(1, 2, 64, 256)
    assert np.array_equal(a, exp1_draws(1, 2, 64, 256))
    for other in (exp1_draws(1, 3, 64, 256), exp1_draws(2, 2, 64, 256)):
        c = np.corrcoef(a.ravel(), other.ravel())[0, 1]
        assert abs(c) < 5.0 / np.sqrt(a.size)
    # the arg-max of i.i.d. draws is uniform: what makes the Gumbel arg-max an unbiased categorical sample
    e = exp1_draws(7, 8, 1 << 15, 64)
    counts = np.bincount(e.argmin(axis=1), minlength=64)
    sigma = np.sqrt(e.shape[0] / 64.0 * (1 - 1 / 64.0))
    assert np.abs(counts - e.shape[0] / 64.0).max() < 5.0 * sigma


def test_keep_mask_rate_and_independence():
    n = 1 << 20
    for p in (0.1, 0.5):
        thr = int(round(p * 65536))
        k = keep_mask(2024, 17, n, thr)
        rate = 1.0 - thr / 65536.0
        assert abs(k.mean() - rate) < 5.0 * np.sqrt(rate * (1 - rate) / n)
        # the two halves of one hash word decide neighbouring elements: they must not agree more than chance
        even, odd = k[0::2].astype(np.float64), k[1::2].astype(np.float64)
        cov = (even * odd).mean() - even.mean() * odd.mean()
        assert abs(cov) < 5.0 * rate * (1 - rate) / np.sqrt(n / 2)
        other = keep_mask(2025, 17, n, thr)
        assert abs((k == other).mean() - (rate * rate + (1 - rate) ** 2)) < 5.0 / np.sqrt(n)
    assert keep_mask(1, 2, 64, 0).all()
