"""CPU: the two GELU forms of the GEMM epilogue (paella_amd/csrc/gemm_device.h) evaluated from the constants AS WRITTEN in the source, with fp32 roundings, against
float64 GELU(erf) (reference: nn.GELU() in the reference's ResBlock / FeedForwardBlock, src/modules.py).
  gelu_erf  -- the exact fp32 path: must be as accurate as 0.5 x (1 + erff(x / sqrt 2)) evaluated in fp32 with a correctly rounded erff;
  gelu_fast -- the opt-in bf16 mode's 12-instruction form: max error 1.27e-4 as the source states.
tools/fit_gelu.py regenerates the constants; the GPU side is tests/test_gpu_fastmode.py::test_bf16_gemm_epilogue_gelu_is_the_12_instruction_fit_within_its_stated_bound."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "paella_amd", "csrc", "gemm_device.h")).read()
F32 = np.float32


def _body(name):
    m = re.search(r"float %s\(float x\) \{(.*?)\n\}" % name, SRC, re.S)
    assert m, name
    return m.group(1)


def _horner(body, var):
    """The Horner chain `float v = c_n; v = fmaf(v, a, c_{n-1}); ...` as written -> coefficients, highest power first."""
    first = re.search(r"float %s = (-?[\d.]+e[+-]\d+)f;" % var, body)
    rest = re.findall(r"%s = fmaf\(%s, a, (-?[\d.]+e[+-]\d+)f\);" % (var, var), body)
    assert first and len(rest) >= 5
    return [F32(first.group(1))] + [F32(v) for v in rest]


def _eval(coefs, a):
    v = np.full_like(a, coefs[0])
    for c in coefs[1:]:
        v = (v.astype(np.float64) * a + c).astype(F32)     # one fp32 FMA
    return v


X = np.linspace(-9, 9, 1800001)
REF = 0.5 * X * (1 + erf(X / np.sqrt(2)))


def test_exact_path_gelu_is_as_accurate_as_fp32_erff():
    body = _body("gelu_erf")
    assert "erff" not in body and "3.92f" in body and "0.70710678118654752440f" in body
    x = X.astype(F32)
    a = np.minimum(np.abs(x) * F32(0.70710678118654752440), F32(3.92)).astype(F32)
    p = (_eval(_horner(body, "q"), a).astype(np.float64) * a).astype(F32)
    e = np.copysign((F32(1) - np.exp2(-p.astype(np.float64)).astype(F32)).astype(F32), x)
    h = (F32(0.5) * x).astype(F32)
    got = (h.astype(np.float64) * e + h).astype(F32)
    libm = (F32(0.5) * x * (F32(1) + erf((x * F32(0.70710678118654752440)).astype(np.float64)).astype(F32))).astype(F32)
    err, err_libm = np.abs(got - REF), np.abs(libm - REF)
    assert err.max() <= 7e-7 and err.max() <= err_libm.max() * 1.05
    inner = np.abs(X) <= 3
    assert err[inner].max() <= 3.5e-7 and np.sqrt((err[inner] ** 2).mean()) <= np.sqrt((err_libm[inner] ** 2).mean()) * 1.05
    assert got[X >= 5.6].tolist() == x[X >= 5.6].tolist()          # erf saturates to exactly 1: gelu(x) = x
    assert np.all(np.abs(got[X <= -5.6]) <= 3e-7)


def test_fast_mode_gelu_meets_its_stated_bound():
    body = _body("gelu_fast")
    x = X.astype(F32)
    a = np.abs(x)
    d = np.maximum(F32(4) - a, F32(0))
    r = _eval(_horner(body, "r"), a)
    got = (np.maximum(x, 0).astype(np.float64) - (a * d).astype(F32).astype(np.float64) * r).astype(F32)
    err = np.abs(got - REF)
    assert 1.0e-4 < err.max() <= 1.3e-4                              # the source says 1.27e-4
    assert np.array_equal(got[np.abs(X) >= 4], np.maximum(x, 0)[np.abs(X) >= 4])
