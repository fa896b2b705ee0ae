"""se3d_div_count (csrc/se3d_stage1.hip): the pair-table entry `cost / n` (S3D:365) and the sweep's `total / nused` (S3D:616-626) are formed
on the device as q0 + (acc - n q0) y with y = RN(1 / n) instead of an fp64 division.  Check on the CPU, with exact rational arithmetic standing in for the FMAs, that the three
operations give the IEEE quotient for every joint count and for accumulators of the kind the kernels produce."""
from fractions import Fraction

import numpy as np


def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # one rounding: float(Fraction) rounds to nearest even


def _div_count(acc, n):
    y = 1.0 / n
    q0 = acc * y
    if not np.isfinite(q0):   # the device returns q0 for a non-finite accumulator (inf - inf in the residual would make it NaN)
        return q0
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


def test_count_divide_for_the_sweeps_view_counts_and_near_rounding_boundaries():
    """Every divisor the one-wavefront sweep can meet (view counts up to 2 x 64 cameras; a few larger ones) on sums of pair costs, and on the
    hardest accumulators: n x (a midpoint between two doubles), rounded — quotients as close to a rounding boundary as a double allows."""
    rng = np.random.default_rng(9)
    for n in list(range(1, 130)) + [255, 256, 257, 511, 512, 513]:
        costs = rng.random((300, 16)) * 0.09
        k = rng.integers(1, 17, size=300)
        acc = [float(costs[i, :k[i]].sum()) for i in range(300)] + [1e6 * j for j in range(1, 4)]
        q = np.ldexp(1.0 + rng.random(300), rng.integers(-40, 40, size=300))
        for x in q:   # a = RN(n (x + ulp(x) / 2)) and its neighbours
            mid = Fraction(float(x)) + Fraction(float(np.spacing(x))) / 2
            a0 = float(mid * n)
            acc += [a0, float(np.nextafter(a0, np.inf)), float(np.nextafter(a0, -np.inf))]
        for a in acc:
            assert _div_count(a, n) == a / n, (a, n)


def test_count_divide_keeps_infinite_and_huge_accumulators():
    """A degenerate epipolar line (norm 0) makes a distance, hence the accumulator, infinite: IEEE division keeps inf, and so must the device's
    three-operation form (the unguarded residual gave NaN); NaN stays NaN; accumulators next to the overflow threshold divide exactly."""
    for n in (1, 2, 3, 7, 16, 17, 64):
        assert _div_count(float("inf"), n) == float("inf")
        assert np.isnan(_div_count(float("nan"), n))
        for a in (1.7e308, 8.9e307, 1e300, 2.0 ** 1023):
            assert _div_count(a, n) == a / n, (a, n)
