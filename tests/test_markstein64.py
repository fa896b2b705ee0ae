"""se3d_div_count (csrc/se3d_stage1.hip): the pair-table entry `cost / n` (S3D:365) is formed on the device as q0 + (acc - n q0) y with
y = RN(1 / n) instead of an fp64 division.  Check on the CPU, with exact rational arithmetic standing in for the FMAs, that the three
operations give the IEEE quotient for every joint count and for accumulators of the kind the kernels produce."""
from fractions import Fraction

import numpy as np


def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # one rounding: float(Fraction) rounds to nearest even


def _div_count(acc, n):
    y = 1.0 / n
    q0 = acc * y
    r = _fma(-float(n), q0, acc)
    return _fma(r, y, q0)


def test_count_divide_is_the_ieee_quotient():
    rng = np.random.default_rng(5)
    for n in range(1, 18):
        # sums of up to 17 float32 distances, like the accumulator; plus arbitrary doubles over a wide exponent range
        k = rng.integers(1, 18, size=10_000)
        terms = rng.random((10_000, 17), dtype=np.float32) * np.float32(0.3)
        acc = np.array([terms[i, :k[i]].astype(np.float64).sum() for i in range(10_000)])
        wide = np.ldexp(1.0 + rng.random(10_000), rng.integers(-60, 60, size=10_000))
        for a in np.concatenate([acc, wide, [0.0]]):
            a = float(a)
            assert _div_count(a, n) == a / n, (a, n)
