"""Coefficients of the RT kernels' exp() interpolant (bart_amd/csrc/kernels.hpp,
exp_rt_n): exp(r) on [-ln2/2, ln2/2] as 1 + r + r^2 g(r), g the polynomial
through the Chebyshev nodes of (exp(r) - 1 - r) / r^2 -- constant and linear
coefficients exactly 1, so optically thin columns (arguments near 0) keep full
precision in the differences of transmittances the intensity sums add up.

    python tools/gen_exp_coef.py [degree ...]

Prints, per degree, the coefficients (highest first) as rounded to double and
their worst relative error in the value and in the derivative against 60-digit
arithmetic.  The kernel's range reduction is one FMA with ln2 rounded to double
(n * ln2 is exact inside the FMA; the constant is off by 2.3e-17 per unit of n)."""
import sys

import mpmath as mp

mp.mp.dps = 60
HALF = mp.log(2) / 2


def g(r):
    r = mp.mpf(r)
    if abs(r) < mp.mpf("1e-12"):
        return mp.mpf(1) / 2 + r / 6
    return (mp.e ** r - 1 - r) / r ** 2


def constrained(deg):
    n = deg - 1
    xs = [HALF * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A, b = mp.matrix(n, n), mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        b[i] = g(x)
    c = mp.lu_solve(A, b)
    return [1.0, 1.0] + [float(c[j]) for j in range(n)]


def errors(c64):
    worst_v = worst_d = mp.mpf(0)
    for i in range(-2000, 2001):
        r = HALF * mp.mpf(i) / 2000
        p = dp = mp.mpf(0)
        for cj in reversed(c64):
            p = p * r + mp.mpf(cj)
        for j in range(len(c64) - 1, 0, -1):
            dp = dp * r + j * mp.mpf(c64[j])
        e = mp.e ** r
        worst_v = max(worst_v, abs(p / e - 1))
        worst_d = max(worst_d, abs(dp / e - 1))
    return float(worst_v), float(worst_d)


def main():
    for d in [int(a) for a in sys.argv[1:]] or [8, 9, 10]:
        c = constrained(d)
        ev, ed = errors(c)
        print("degree %d: max rel err %.2e (value), %.2e (derivative)" % (d, ev, ed))
        print("  " + ", ".join("%.17g" % x for x in c[::-1]))
    ln2d = float(mp.log(2))
    print("ln2 (double) = %.20e, error per unit n = %.2e" % (ln2d, abs(float(mp.log(2) - mp.mpf(ln2d)))))


if __name__ == "__main__":
    main()
