#!/usr/bin/env python
"""Study for round 5 (CPU only): how cheap can the erf of the GEGLU get before the fp16 rounding that follows it notices?

The GEGLU runs beside the MFMAs, where VALU time adds to matrix time (HISTORY 5.2d): 13 VALU per element (two transcendental) is ~45 % of the
fused feed-forward's iteration and +20..37 % on the matrix time of the C = 640 / 1280 first-linear GEMMs.  Candidates, all in fp32 arithmetic
as the kernel would run them (numpy, every intermediate rounded to fp32), against erf in fp64:
  as26    Abramowitz-Stegun 7.1.26 (shipped): t = 1/(1+pz), 5-term polynomial, exp(-z^2)           13 VALU (2 transcendental)
  as25    Abramowitz-Stegun 7.1.25: 3-term polynomial                                                11 VALU (2 transcendental)
  poly    erf(z) ~ clamp(z * P(z^2)) with an odd least-squares polynomial on |z| <= 3.2, degree 2n+1  n+3 VALU (0 transcendental)
Reported: max |gelu error|, the same in units of the fp16 ulp at the value, and the share of a * gelu(g) products (a, g ~ N(0, 1.5),
the feed-forward's pre-activation scale at the 64^2 level) whose fp16 rounding differs from the exact one's."""
import numpy as np
from scipy.special import erf

f32 = np.float32
SQ = f32(0.70710678118654752440)


def gelu_exact(x):
    return 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))


def as26(x):
    ax = np.abs(x)
    t = (f32(1) / (ax * f32(0.3275911 * 0.70710678118654752440) + f32(1))).astype(f32)
    zs = (ax * f32(0.84932180028801904272)).astype(f32)
    e = np.exp2(-(zs * zs)).astype(f32)
    q = (f32(1.061405429) * t + f32(-1.453152027)).astype(f32)
    for c in (1.421413741, -0.284496736, 0.254829592):
        q = (q * t + f32(c)).astype(f32)
    q = (q * t).astype(f32)
    return (f32(0.5) * (ax * (f32(1) - q * e) + x)).astype(f32)


def as25(x):
    ax = np.abs(x)
    t = (f32(1) / (ax * f32(0.47047 * 0.70710678118654752440) + f32(1))).astype(f32)
    zs = (ax * f32(0.84932180028801904272)).astype(f32)
    e = np.exp2(-(zs * zs)).astype(f32)
    q = (f32(0.7478556) * t + f32(-0.0958798)).astype(f32)
    q = (q * t + f32(0.3480242)).astype(f32)
    q = (q * t).astype(f32)
    return (f32(0.5) * (ax * (f32(1) - q * e) + x)).astype(f32)


def make_poly(n_terms, zmax=3.2):
    z = np.linspace(0, zmax, 20001)
    A = np.stack([z ** (2 * k + 1) for k in range(n_terms)], 1)
    w = 1.0 / np.maximum(erf(z), 1e-3)              # relative weighting near 0
    coef, *_ = np.linalg.lstsq(A * w[:, None], erf(z) * w, rcond=None)
    return coef.astype(f32)


def poly(x, coef, zmax=3.2):
    z = np.clip((x * SQ).astype(f32), f32(-zmax), f32(zmax))
    z2 = (z * z).astype(f32)
    p = np.full_like(z, coef[-1])
    for c in coef[-2::-1]:
        p = (p * z2 + c).astype(f32)
    e = np.clip((p * z).astype(f32), f32(-1), f32(1))
    return (f32(0.5) * x * (f32(1) + e)).astype(f32)


def report(name, fn, valu):
    x = np.linspace(-8, 8, 4_000_001).astype(f32)
    ref = gelu_exact(x)
    got = fn(x).astype(np.float64)
    err = np.abs(got - ref)
    ulp = np.spacing(np.abs(ref).astype(np.float16)).astype(np.float64)
    rng = np.random.default_rng(0)
    a, g = (rng.standard_normal(2_000_000) * 1.5).astype(f32), (rng.standard_normal(2_000_000) * 1.5).astype(f32)
    exact16 = (a.astype(np.float64) * gelu_exact(g)).astype(np.float16)
    got16 = (a * fn(g)).astype(np.float16)
    print(f"{name:10s} {valu:>22s}  max |err| {err.max():.2e}   max err / fp16 ulp {np.max(err / ulp):7.3f}   fp16 products that differ {np.mean(exact16 != got16) * 100:6.3f} %")


if __name__ == "__main__":
    report("as26", as26, "13 (2 transcendental)")
    report("as25", as25, "11 (2 transcendental)")
    for n in (4, 5, 6, 7):
        c = make_poly(n)
        report(f"poly{2 * n - 1}", lambda v, c=c: poly(v, c), f"{n + 4} (0 transcendental)")
