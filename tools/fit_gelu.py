"""Fits and checks the two GELU forms of gemm_device.h (CPU only: numpy + scipy).
  gelu_erf  (exact fp32 path): erf(a) = sign (1 - 2^-P(|a|)), P = degree-8 weighted minimax fit of -log2 erfc on [0, 3.92]; evaluated here with fp32 roundings
                                and compared with 0.5 x (1 + erff(x / sqrt 2)) evaluated in fp32 (what libm's erff gives at best).
  gelu_fast (opt-in bf16 mode): gelu(x) = max(x, 0) - |x| (4 - |x|)+ R(|x|), R = degree-6 fit; max error 1.27e-4.
Usage: python tools/fit_gelu.py   (prints the coefficients and the error figures quoted in the source)"""
import numpy as np
from scipy.special import erf, erfc


def lawson(V, f, w, iters=300):
    lw = np.ones_like(f)
    for _ in range(iters):
        W = w * lw
        c, *_ = np.linalg.lstsq(V * W[:, None], f * W, rcond=None)
        e = np.abs((V @ c - f) * w)
        lw = lw * (e / e.mean() + 1e-9) ** 0.5
        lw /= lw.mean()
    return c


def to_monomial(c, X):
    return np.polynomial.Polynomial(np.polynomial.chebyshev.cheb2poly(c))(np.polynomial.Polynomial([-1, 2 / X])).coef


def horner32(coef, a):
    q = np.full_like(a, np.float32(coef[-1]))
    for cc in coef[-2::-1]:
        q = (q.astype(np.float64) * a + np.float32(cc)).astype(np.float32)   # one fp32 FMA
    return q


def fit_exact(A=3.92, n=8):
    k = np.arange(20000)
    a = (np.cos(np.pi * (k + 0.5) / 20000) + 1) / 2 * A
    a = a[a > 1e-4]
    f = -np.log2(erfc(a)) / a
    w = erfc(a) * np.log(2) * a * (0.5 * a * np.sqrt(2) + 0.2)   # the error a coefficient error causes in gelu
    u = a / A * 2 - 1
    return to_monomial(lawson(np.polynomial.chebyshev.chebvander(u, n - 1), f, w), A).astype(np.float32)


def fit_fast(X=4.0, n=7):
    k = np.arange(8000)
    a = (np.cos(np.pi * (k + 0.5) / 8000) + 1) / 2 * X
    f = 0.5 * erfc(a / np.sqrt(2)) / (X - a)
    u = a / X * 2 - 1
    return to_monomial(lawson(np.polynomial.chebyshev.chebvander(u, n - 1), f, a * (X - a), 400), X).astype(np.float32)


if __name__ == "__main__":
    xx = np.linspace(-8, 8, 4000001)
    ref = 0.5 * xx * (1 + erf(xx / np.sqrt(2)))
    x = xx.astype(np.float32)
    cur = (np.float32(0.5) * x * (np.float32(1) + erf((x * np.float32(0.70710678118654752440)).astype(np.float64)).astype(np.float32))).astype(np.float32)
    print("0.5 x (1 + erff(x / sqrt 2)) in fp32, erff correctly rounded: max |error| %.3e" % np.abs(cur - ref).max())
    c = fit_exact()
    a = np.minimum(np.abs(x) * np.float32(0.70710678118654752440), np.float32(3.92)).astype(np.float32)
    p = (horner32(c, a).astype(np.float64) * a).astype(np.float32)
    e = np.copysign((np.float32(1) - np.exp2(-p.astype(np.float64)).astype(np.float32)).astype(np.float32), x)
    h = (np.float32(0.5) * x).astype(np.float32)
    g = (h.astype(np.float64) * e + h).astype(np.float32)
    print("gelu_erf coefficients (a^1 .. a^8):", ", ".join("%.9ef" % v for v in c))
    print("gelu_erf: max |error| %.3e" % np.abs(g - ref).max())
    c = fit_fast()
    a = np.abs(x)
    d = np.maximum(np.float32(4) - a, np.float32(0))
    gf = (np.maximum(x, 0).astype(np.float64) - (a * d).astype(np.float32).astype(np.float64) * horner32(c, a)).astype(np.float32)
    print("gelu_fast coefficients (a^0 .. a^6):", ", ".join("%.9ef" % v for v in c))
    print("gelu_fast: max |error| %.3e" % np.abs(gf - ref).max())
