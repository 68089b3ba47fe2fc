"""Fit of the bf16-mode GELU form of csrc/preattn_act.hip:  Phi(x) ~= sigmoid(x (p0 + p1 x^2 + p2 x^4))  on [-6, 6]
(iteratively re-weighted least squares towards the minimax fit of x*Phi(x)); prints the coefficients and the maximum absolute errors of
the value and of the derivative of the approximation against the exact erf forms."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf

x = np.linspace(-6, 6, 24001)
phi = 0.5 * (1 + erf(x / np.sqrt(2)))
ge = x * phi
de = phi + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


def sig(p, x):
    x2 = x * x
    return 1 / (1 + np.exp(-x * (p[0] + x2 * (p[1] + x2 * p[2]))))


p, w = np.array([1.5957691216, 0.0713548162726, 0.0]), np.ones_like(x)
for _ in range(60):
    p = least_squares(lambda q: w * (x * sig(q, x) - ge), p, xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    e = np.abs(x * sig(p, x) - ge)
    w = w * (1 + 4 * e / e.max())
    w /= w.mean()
s, x2 = sig(p, x), x * x
dt = s + x * s * (1 - s) * (p[0] + 3 * p[1] * x2 + 5 * p[2] * x2 * x2)
print("p =", p, " max |value err| = %.2e  max |derivative err| = %.2e" % (np.abs(x * s - ge).max(), np.abs(dt - de).max()))
