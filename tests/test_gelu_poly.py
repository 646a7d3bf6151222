"""CPU: the GELU polynomial the HIP kernels carry (csrc/common.hpp kGeluC; tools/gelu_fit.py derives it) evaluated as the kernels evaluate it -- float32, one
rounding per fma, Horner in t = 2 w^2 / 25 - 1 -- against scipy's erf: the accuracy the header claims, exact saturation, oddness."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coefs():
    src = open(os.path.join(ROOT, "realcamnet_amd", "csrc", "common.hpp")).read()
    body = re.search(r"kGeluC\[13\]\s*=\s*\{([^}]*)\}", src).group(1)
    c = [np.float32(x.strip().rstrip("f")) for x in body.split(",")]
    assert len(c) == 13
    return c


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _gelu32(v, c):
    w = np.clip(v, np.float32(-5), np.float32(5)).astype(np.float32)
    t = _fma((w * w).astype(np.float32), np.full_like(w, np.float32(0.08)), np.full_like(w, np.float32(-1)))
    q = np.full_like(w, c[12])
    for k in range(11, -1, -1):
        q = _fma(q, t, np.full_like(w, c[k]))
    e = (w * q).astype(np.float32)
    hv = (v * np.float32(0.5)).astype(np.float32)
    return _fma(hv, e, hv), e


def test_gelu_polynomial_matches_erf_and_saturates():
    c = _coefs()
    v = np.linspace(-9, 9, 1_000_001).astype(np.float32)
    g, e = _gelu32(v, c)
    ref_e = erf(v.astype(np.float64) / np.sqrt(2.0))
    ref_g = 0.5 * v.astype(np.float64) * (1.0 + ref_e)
    assert np.abs(e - ref_e).max() < 1e-6                      # header: 6.7e-7 (the Abramowitz-Stegun 7.1.28 form it replaced: 1.9e-6 in float32)
    assert np.abs(g - ref_g).max() < 3e-6
    assert np.abs(e).max() <= 1.0 and (g[v <= -5] == 0).all() and (g[v >= 5] == v[v >= 5]).all()
    _, en = _gelu32(-v, c)
    assert np.array_equal(en, -e)                              # the erf part is odd, bit for bit
