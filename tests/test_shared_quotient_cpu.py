"""The large-rig pair kernel's quotient (csrc/se3d_stage1.hip: se3d_div_shared on the shared reciprocal se3d_safe_rcp) restated with exact float32
arithmetic on the CPU (tests/float32_emul.py) against IEEE division: calcCost's `abs(p2.dot(l1)) / l1.head(2).norm()`, S3D:358-361, must come out bit
for bit.  The device runs the same check on 2^33 random pairs (tests/test_gpu_stage1.py); this file covers what a random draw does not reach — all
numerators of a binade for divisors at the edges of the guard, quotients constructed to lie 2^-25 .. 2^-22 ulp from a rounding boundary — and shows that
the check can fail (a reciprocal one ulp off does)."""
import numpy as np

from float32_emul import div_shared, safe_rcp


def _mismatches(a, b, y=None):
    y = safe_rcp(b) if y is None else y
    ok = y != 0
    q = div_shared(a, b, y)
    ref = (a / b).astype(np.float32)
    return int((ok & (q.view(np.uint32) != ref.view(np.uint32))).sum()), int(ok.sum())


def _random_pairs(rng, n):
    b = rng.integers(0x30800000, 0x4e800000, n, dtype=np.uint32).view(np.float32)              # [2^-30, 2^30)
    za = rng.integers(0, 2 ** 32, n, dtype=np.uint64)
    wide = (0x16000000 + za % (0x5d800000 - 0x16000000 + 1)).astype(np.uint32).view(np.float32)   # [2^-83, 2^60]
    eb = (b.view(np.uint32) & np.uint32(0x7f800000)).astype(np.uint64)
    near = (eb - (12 << 23) + za % (24 << 23)).astype(np.uint32).view(np.float32)
    a = np.where(np.arange(n) % 3 == 0, near, wide).astype(np.float32)
    a[63::64] = 0
    return a, b


def _near_midpoint_pairs(divisors):
    """(A, B) with A / B = M 2^-24 - s / (2^24 B) for a 25-bit odd M and s = +-1 .. +-6: the quotient lies |s| 2^-25 (2^24 / B) ulp from the rounding
    boundary M 2^-24 — the closest a float32 quotient can come (|A - B M 2^-24| is a multiple of 2^-24)."""
    B = np.asarray(divisors, np.int64)
    inv = B.copy()
    for _ in range(6):
        inv = (inv * (2 - B * inv)) & ((1 << 24) - 1)      # B^-1 mod 2^24 (B odd)
    As, Bs = [], []
    for s in range(-6, 7):
        if s == 0:
            continue
        M = (1 << 24) + ((s * inv) & ((1 << 24) - 1))
        A = (B * M - s) >> 24
        a = A.astype(np.float32)
        ok = ((M & 1) == 1) & (a.astype(np.int64) == A)
        As.append(a[ok]); Bs.append(B[ok].astype(np.float32))
    return np.concatenate(As), np.concatenate(Bs)


def test_quotient_is_the_ieee_quotient_on_random_operands():
    rng = np.random.default_rng(5)
    bad = tot = 0
    for _ in range(4):
        m, n = _mismatches(*_random_pairs(rng, 1_000_000))
        bad, tot = bad + m, tot + n
    assert bad == 0 and tot > 3_900_000, (bad, tot)


def test_quotient_next_to_rounding_boundaries():
    """Divisors just below 2^24 ulps (the reciprocal's rounding error is at its maximum for odd 2^24 - k) and random odd divisors."""
    rng = np.random.default_rng(9)
    for div in ((1 << 24) - np.arange(3, 1 << 20, 2, dtype=np.int64), rng.integers(1 << 23, 1 << 24, 1_000_000, dtype=np.int64) | 1):
        a, b = _near_midpoint_pairs(div)
        bad, tot = _mismatches(a, b)
        assert bad == 0 and tot > 500_000, (bad, tot)


def test_quotient_on_every_numerator_of_a_binade():
    """All 2^23 numerators of [2^23, 2^24) against divisors at the edges of the guard (2^24 - 3, 2^24 - 2, 2^23 + 1), sqrt(2) 2^23, 2/3 2^24."""
    A = np.arange(1 << 23, 1 << 24, dtype=np.int64).astype(np.float32)
    for B in ((1 << 24) - 3, (1 << 24) - 2, (1 << 23) + 1, 0xB504F3, 0xAAAAAB):
        bad, tot = _mismatches(A, np.full(len(A), B, np.float32))
        assert bad == 0 and tot == len(A), (B, bad)


def test_zero_tiny_and_exact_quotients():
    rng = np.random.default_rng(2)
    n = 300_000
    b = rng.integers(0x30800000, 0x4e800000, n, dtype=np.uint32).view(np.float32)
    exact = (b.astype(np.float64) * rng.integers(1, 1 << 12, n)).astype(np.float32)
    tiny = rng.integers(0x16000000, 0x17000000, n, dtype=np.uint32).view(np.float32)        # 2^-83 ..: residuals in the denormal range
    for a in (np.zeros(n, np.float32), exact, tiny):
        bad, tot = _mismatches(a, b)
        assert bad == 0 and tot > 0, (bad, tot)


def test_the_check_can_fail_a_reciprocal_one_ulp_off_is_not_enough():
    """Negative control: the same three instructions on y = the float NEXT to RN(1 / b) miss the correctly rounded quotient on near-boundary operands
    (why the kernel stages the IEEE reciprocal, not v_rcp_f32's)."""
    a, b = _near_midpoint_pairs(np.random.default_rng(9).integers(1 << 23, 1 << 24, 1_000_000, dtype=np.int64) | 1)
    y = np.nextafter(safe_rcp(b), np.float32(np.inf)).astype(np.float32)
    bad, tot = _mismatches(a, b, y)
    assert bad > 0, tot
