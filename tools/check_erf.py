"""Numerical check (CPU, numpy fp32 with fused multiply-adds emulated in double) of the branch-free erf used by every GELU of the path
(acai_omr_amd/csrc/common.h: acai_erff): maximum absolute / relative error against math.erf over [-6, 6]."""
import math
import numpy as np

f = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f)


def acai_erff(a):
    a = a.astype(f)
    t, s = np.abs(a), a * a
    c = lambda v: np.full_like(a, v)
    r = fma(c(-1.72853470e-5), t, c(3.83197126e-4))
    u = fma(c(-3.88396438e-3), t, c(2.42546219e-2))
    r = fma(r, s, u)
    for k in (-1.06777877e-1, -6.34846687e-1, -1.28717512e-1):
        r = fma(r, t, c(k))
    r = fma(r, t, -t)
    r = (f(1.0) - np.exp2(r.astype(np.float64) * 1.4426950408889634).astype(f)).astype(f)
    big = np.copysign(r, a)
    q = c(-5.96761703e-4)
    for k in (4.99119423e-3, -2.67681349e-2, 1.12819925e-1, -3.76125336e-1, 1.28379166e-1):
        q = fma(q, s, c(k))
    small = fma(q, a, a)
    return np.where(t > f(0.927734375), big, small)


if __name__ == "__main__":
    x = np.linspace(-6, 6, 400001).astype(f)
    ref = np.array([math.erf(float(v)) for v in x])
    err = np.abs(acai_erff(x).astype(np.float64) - ref)
    print("max abs err", err.max(), "max rel err", (err / np.maximum(np.abs(ref), 1e-30))[np.abs(x) > 1e-3].max())
    assert err.max() < 1e-7
