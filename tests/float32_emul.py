"""Exact float32 arithmetic of the device sequences on the CPU (numpy): a fused multiply-add of float32 operands, correctly rounded — the product is
exact in float64, the sum is rounded to odd in float64 (Boldo & Melquiond: 53 >= 2 x 24 + 2 bits make the second rounding innocuous) and then to
float32 — and the pair kernels' quotient sequence built on it (csrc/se3d_stage1.hip).  Test infrastructure only."""
import numpy as np


def fma32(a, b, c):
    a64, b64, c64 = (np.asarray(v, np.float32).astype(np.float64) for v in (a, b, c))
    p = a64 * b64                                   # exact: 48 significant bits
    s = p + c64
    bb = s - p
    e = (p - (s - bb)) + (c64 - bb)                 # TwoSum: s + e = p + c exactly
    even = (np.atleast_1d(s).view(np.uint64) & np.uint64(1)) == 0
    fix = (e != 0) & even.reshape(np.shape(s)) & np.isfinite(s)
    s = np.where(fix, np.nextafter(s, np.where(e > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


def safe_rcp(b):
    """se3d_safe_rcp: RN(1 / b) for divisors in [2^-30, 2^30] whose significand is not all ones, else 0."""
    b = np.asarray(b, np.float32)
    u = b.view(np.uint32)
    ex = (u >> np.uint32(23)) & np.uint32(0xff)
    ok = (ex >= 97) & (ex <= 157) & ((u & np.uint32(0x7fffff)) != np.uint32(0x7fffff))
    with np.errstate(divide="ignore"):
        return np.where(ok, np.float32(1.0) / b, np.float32(0.0)).astype(np.float32)


def div_shared(a, b, y):
    """se3d_div_shared: q0 = RN(a y); r = fma(-b, q0, a); q = fma(r, y, q0)."""
    a, b, y = (np.asarray(v, np.float32) for v in (a, b, y))
    q0 = (a * y).astype(np.float32)
    return fma32(fma32(-b, q0, a), y, q0)
