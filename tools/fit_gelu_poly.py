#!/usr/bin/env python3
"""Minimax polynomial fits of the bf16-mode GELU of csrc/preattn_act.hip (round 5):
    GELU(x)  - x/2 = x^2 P(x^2)        GELU'(x) - 1/2 = x R(x^2)        on [-C, C], P and R of degree N-1 in x^2
by Lawson's iteratively re-weighted least squares on the ABSOLUTE error; prints the coefficients (lowest order first) and the maximum
absolute errors against the exact erf forms, evaluated in fp32 Horner form as the kernel does."""
import sys

import numpy as np
from scipy.special import erf

C = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def lawson(A, y, iters=300):
    w = np.ones(len(y)) / len(y)
    for _ in range(iters):
        W = np.sqrt(w)
        coef, *_ = np.linalg.lstsq(A * W[:, None], y * W, rcond=None)
        e = np.abs(A @ coef - y)
        w = w * (e + 1e-12)
        w /= w.sum()
    return coef


x = np.linspace(1e-6, C, 40001)
Phi = 0.5 * (1 + erf(x / np.sqrt(2)))
pdf = np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
P = lawson(np.stack([x ** (2 * k + 2) for k in range(N)], 1), x * Phi - 0.5 * x)
R = lawson(np.stack([x ** (2 * k + 1) for k in range(N)], 1), Phi + x * pdf - 0.5)


def horner32(c, x2):
    p = np.float32(c[-1]) * x2 + np.float32(c[-2])
    for k in range(len(c) - 3, -1, -1):
        p = p * x2 + np.float32(c[k])
    return p


xx = np.linspace(-2 * C, 2 * C, 400001).astype(np.float32)
xc = np.clip(xx, -C, C).astype(np.float32)
g = xx * (xc * horner32(P, xc * xc) + np.float32(0.5))
dg = xc * horner32(R, xc * xc) + np.float32(0.5)
x64 = xx.astype(np.float64)
Phi64 = 0.5 * (1 + erf(x64 / np.sqrt(2)))
print("range +-%g, %d terms" % (C, N))
print("P:", ", ".join("%.9e" % v for v in P))
print("R:", ", ".join("%.9e" % v for v in R))
print("max |GELU err| on [-2C, 2C] (fp32 Horner, clamped argument): %.2e" % np.abs(g - x64 * Phi64).max())
print("max |GELU' err|: %.2e" % np.abs(dg - (Phi64 + x64 * np.exp(-x64 * x64 / 2) / np.sqrt(2 * np.pi))).max())
