"""Numerical check (CPU, numpy fp32 with fused multiply-adds emulated in double) of the branch-free GELU / GELU' used by every kernel of the path
(acai_omr_amd/csrc/common.h: acai_half_erfc_abs, gelu_erf, gelu_erf_grad) against the exact functions (mpmath): maximum absolute error over a
dense fp32 grid, and - the number that matters for the autocast path - how many of ALL finite bf16 inputs give a bf16-rounded GELU that differs
from the correctly rounded one."""
import numpy as np

f = np.float32
COEF = [-1.627925070e+00, -9.181654744e-01, -1.496994187e-01, 3.089617088e-02, -3.664269981e-03, 1.420412202e-04]


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f)


def half_erfc_abs(x):
    x = x.astype(f)
    t = np.minimum((np.abs(x) * f(0.70710678118654752440)).astype(f), f(12.0))
    q = np.full_like(x, f(COEF[-1]))
    for c in COEF[-2::-1]:
        q = fma(q, t, np.full_like(x, f(c)))
    p = fma(q, t, np.full_like(x, f(-1.0)))
    return np.exp2(p.astype(np.float64)).astype(f)


def gelu(x):
    x = x.astype(f)
    return fma(-np.abs(x), half_erfc_abs(x), np.maximum(x, f(0)))


def gelu_grad(x):
    x = x.astype(f)
    h = half_erfc_abs(x)
    phi = np.where(x < 0, h, (f(1.0) - h).astype(f))
    e = np.exp2(fma((x * x).astype(f), np.full_like(x, f(-0.72134752044448170368)), np.full_like(x, f(-1.3257480647361593))).astype(np.float64)).astype(f)
    return fma(x, e, phi)


def round_bf16(v):
    b = v.astype(f).view(np.uint32).astype(np.uint64)
    return ((b + 0x7FFF + ((b >> 16) & 1)) >> 16 << 16).astype(np.uint32).view(f)


if __name__ == "__main__":
    import mpmath as mp
    mp.mp.dps = 40
    x = np.linspace(-8, 8, 40001).astype(f)
    ref = np.array([float(mp.mpf(float(v)) * mp.ncdf(mp.mpf(float(v)))) for v in x])
    refg = np.array([float(mp.ncdf(mp.mpf(float(v))) + mp.mpf(float(v)) * mp.npdf(mp.mpf(float(v)))) for v in x])
    e1, e2 = np.abs(gelu(x).astype(np.float64) - ref).max(), np.abs(gelu_grad(x).astype(np.float64) - refg).max()
    print("max |gelu error| over [-8, 8]:", e1, "  max |gelu' error|:", e2)
    assert e1 < 1e-6 and e2 < 1e-6
    xb = (np.arange(65536, dtype=np.uint32) << 16).view(f)
    xb = xb[np.isfinite(xb)]
    refb = np.array([float(mp.mpf(float(v)) * mp.ncdf(mp.mpf(float(v)))) for v in xb])
    g = gelu(xb)
    assert np.all(np.isfinite(g)) and np.all(np.abs(g.astype(np.float64) - refb) <= 1e-6 * np.maximum(1.0, np.abs(refb)))   # incl. |x| up to 3e38
    core = np.abs(xb) < 8
    bad = (round_bf16(g) != round_bf16(refb.astype(f))) & core
    print("bf16 inputs in |x| < 8 whose bf16-rounded GELU differs from the correctly rounded one:", int(bad.sum()), "of", int(core.sum()),
          "; largest such input:", float(xb[bad].max()) if bad.any() else None)
    assert bad.sum() < 200 and (not bad.any() or xb[bad].max() < -3.0)
