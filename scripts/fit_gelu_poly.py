#!/usr/bin/env python3
"""Coefficients of pk_common.h's polynomial GELU:  erf(x / sqrt 2) ~ xc * P(xc^2),  xc = clamp(x, -X, X),  deg P = 8.

Weighted least squares on Chebyshev nodes, re-weighted towards the maximum error (a Remez-like iteration); prints the fp32
coefficients and the errors of gelu / gelu' evaluated in fp32 the way the kernel does (Horner with FMAs)."""
import numpy as np
from scipy.special import erf

X, DEG = 4.2, 8
s = (np.cos(np.linspace(0, np.pi, 8001)) * 0.5 + 0.5) * X * X
x = np.sqrt(s)
A = np.stack([x * s ** k for k in range(DEG + 1)], 1)
y = erf(x / np.sqrt(2))
w, best = np.ones_like(x), None
for _ in range(400):
    c, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
    e = np.abs(A @ c - y)
    if best is None or e.max() < best[1]:
        best = (c.copy(), e.max())
    w = w * (1 + 2 * e / e.max())
    w /= w.mean()
c, err = best
print(f"max |x P(x^2) - erf(x / sqrt 2)| on [0, {X}] = {err:.3e};  1 - erf(X / sqrt 2) = {1 - erf(X / np.sqrt(2)):.3e}")
for k, v in enumerate(c):
    print(f"#define PK_GELU_C{k} {np.float32(v)!r}".replace("np.float32(", "").replace(")", "f"))
xx = np.linspace(-8, 8, 800001).astype(np.float32)
xc = np.clip(xx, -np.float32(X), np.float32(X))
ss = (xc * xc).astype(np.float32)
p = np.float32(c[-1]) * np.ones_like(ss)
for k in range(DEG - 1, -1, -1):
    p = (p * ss + np.float32(c[k])).astype(np.float32)
ev = (xc * p).astype(np.float32)
h = np.float32(0.5) * xx
g = (h * ev + h).astype(np.float32)
x64 = xx.astype(np.float64)
print(f"gelu  max abs error {np.abs(g - 0.5 * x64 * (1 + erf(x64 / np.sqrt(2)))).max():.3e}")
gd = (0.5 + 0.5 * ev) + xx * np.float32(0.3989422804) * np.exp(-0.5 * xx * xx)
print(f"gelu' max abs error {np.abs(gd - (0.5 * (1 + erf(x64 / np.sqrt(2))) + x64 * np.exp(-0.5 * x64 ** 2) / np.sqrt(2 * np.pi))).max():.3e}")
