#!/usr/bin/env python3
"""The coefficients of csrc/vit.hip's erf-GELU: gelu(x) = max(x, 0) - |x| 2^-(1 + a q(a)), a = |x|, q of degree 4, i.e. erfc(a / sqrt 2) / 2 ~ 2^-P(a) with
P(a) = 1 + c1 a + ... + c5 a^5 (P(0) = 1 exactly: erfc(0) / 2 = 1 / 2).  Iteratively re-weighted least squares towards the minimax of the GELU's ABSOLUTE error
0.5 a |2^-P~ - erfc| on [0, 8] (beyond, both are < 1e-14); then the float32 evaluation is checked over a dense grid and at the extremes.  CPU only."""
import numpy as np
from scipy.special import erf, erfc

DEG = 5
x = np.linspace(0, 8.0, 400001)
E = erfc(x / np.sqrt(2))
P = -np.log2(np.maximum(E, 1e-300))
A = np.vander(x, DEG + 1, increasing=True)[:, 1:]
sens = 0.5 * x * E * np.log(2) + 1e-12
w = np.ones_like(x)
for _ in range(60):
    c, *_ = np.linalg.lstsq(A * (w * sens)[:, None], P * w * sens, rcond=None)
    err = np.abs(0.5 * x * (2.0 ** (-(A @ c)) - E))
    w = w * (1 + 3 * err / err.max())
    w /= w.mean()
print("coefficients c1..c5:", [f"{v:.8e}" for v in c], " max |gelu error| in exact arithmetic:", err.max())
c32 = c.astype(np.float32)
xs = np.concatenate([np.linspace(-12, 12, 2000001), [0.0, -0.0, 1e-30, -1e-30, 50, -50, 1e4, -1e4, 65504, -65504]]).astype(np.float32)
ax = np.abs(xs)
q = c32[4]
for k in (3, 2, 1, 0):
    q = (q * ax + c32[k]).astype(np.float32)
with np.errstate(over="ignore"):
    g = ((xs + ax) * np.float32(0.5) - ax * np.exp2(-(ax * q + np.float32(1))).astype(np.float32)).astype(np.float32)
ref = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
print("float32 evaluation: max |error|", np.abs(g - ref).max(), "finite everywhere:", bool(np.isfinite(g).all()))
dP = np.diff(np.polyval(np.concatenate([[1.0], c])[::-1], np.linspace(0, 200, 200001)))
print("P monotone on [0, 200]:", bool((dP > 0).all()))
