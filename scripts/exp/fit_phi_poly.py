"""Coefficients of tc_common.h's TC_PHI_*: the normal CDF as 0.5 + t P(t^2), t = clamp(x, -X0, X0), P of degree 8 in t^2, fitted for the
smallest maximum ABSOLUTE error on [0, X0] (Lawson's iteratively reweighted least squares on Chebyshev nodes), then checked in fp32 Horner
arithmetic over [-6, 6] against scipy's erf.   python scripts/exp/fit_phi_poly.py [X0] [degree]"""
import sys
import numpy as np
from scipy.special import erf

X0 = float(sys.argv[1]) if len(sys.argv) > 1 else 4.2
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 4000
x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 * X0 + 0.5 * X0
s = x * x
A = np.stack([x * s ** k for k in range(deg + 1)], 1)
y = 0.5 * erf(x / np.sqrt(2))
w = np.ones(n)
for _ in range(200):
    c = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
    e = np.abs(A @ c - y)
    w = w * (0.5 + e / e.max())
    w /= w.mean()
xx = np.linspace(-6, 6, 600001).astype(np.float32)
t = np.clip(xx, -X0, X0).astype(np.float32)
ss = (t * t).astype(np.float32)
c32 = c.astype(np.float32)
p = np.full_like(xx, c32[deg])
for k in range(deg - 1, -1, -1):
    p = (p * ss + c32[k]).astype(np.float32)
phi = (t * p + np.float32(0.5)).astype(np.float32)
ref = 0.5 * (1 + erf(xx.astype(np.float64) / np.sqrt(2)))
print(f"// X0 = {X0}, degree {deg} in t^2: max |error| {np.abs(phi - ref).max():.3e} over [-6, 6] in fp32 Horner arithmetic; range [{phi.min():.3e}, {phi.max():.8f}]")
print(f"#define TC_PHI_X0 {X0}f")
for k, v in enumerate(c):
    print(f"#define TC_PHI_C{k} {v:.10e}f")
