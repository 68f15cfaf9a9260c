"""Generates bart_amd/csrc/expint_coef.hpp: Chebyshev tables for the exponential
integral E_2(x) used by the device T(p) model of Line et al. (xi(), reference
code/PT.py:722-739, which calls scipy.special.expn(2, x)).

  0 <= x <= 1 :  E_2(x) = x ln x + Q(x),  Q entire:  Q(x) = e^-x + x (gamma - S(x)),
                 S(x) = sum_{k>=1} (-1)^(k+1) x^k / (k k!)
  1 <  x < 1024: E_2(x) = e^-x F(x),  F(x) = e^x E_2(x) fitted per half octave
                 [2^(j/2), 2^((j+1)/2)], j = 0..19 (x > 709.78 returns 0 before that)

Reference values come from 90-digit decimal arithmetic (series for x < 40, Lentz
continued fraction above); coefficients are Chebyshev interpolants at Chebyshev
nodes, truncated where the tail is below 2e-18 of the leading coefficient.  The
script also evaluates the tables exactly as the kernel does (Clenshaw in fp64)
against the reference on a dense grid and prints the worst relative error."""
import os
import sys
from decimal import Decimal as D, getcontext

import numpy as np

getcontext().prec = 90
GAMMA = D("0.5772156649015328606065120900824024310421593359399235988057672348848677267776646709369470632917467495")


def dexp(x):
    return x.exp()


def e1_series(x):
    s, term, k = D(0), D(1), 0
    while True:
        k += 1
        term = term * (-x) / k          # (-x)^k / k!
        add = -term / k                 # (-1)^(k+1) x^k / (k k!)
        s += add
        if abs(add) < D(10) ** -85 and k > x:
            break
    return -GAMMA - x.ln() + s


def e2_cf(x):
    """E_2 by the modified Lentz continued fraction (Numerical Recipes form)."""
    n = 2
    b = x + n
    c = D(10) ** 80
    d = 1 / b
    h = d
    i = 0
    while True:
        i += 1
        a = -D(i) * (n - 1 + i)
        b += 2
        d = 1 / (a * d + b)
        c = b + a / c
        delta = c * d
        h *= delta
        if abs(delta - 1) < D(10) ** -80:
            break
    return h * dexp(-x)


def e2(x):
    x = D(x)
    if x == 0:
        return D(1)
    if x < 40:
        return dexp(-x) - x * e1_series(x)
    return e2_cf(x)


def q_small(x):
    """Q(x) = E_2(x) - x ln x, evaluated without the cancelling logarithms."""
    x = D(x)
    if x == 0:
        return D(1)
    s, term, k = D(0), D(1), 0
    while True:
        k += 1
        term = term * (-x) / k
        add = -term / k
        s += add
        if abs(add) < D(10) ** -85:
            break
    return dexp(-x) + x * (GAMMA - s)


def f_large(x):
    x = D(x)
    return e2(x) * dexp(x)


PI = D("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798")


def dcos(x):
    """cos of a Decimal (Taylor series after reduction to [-pi, pi])."""
    x = x - (x / (2 * PI)).to_integral_value() * 2 * PI
    s, term, k = D(1), D(1), 0
    while abs(term) > D(10) ** -88:
        k += 2
        term = -term * x * x / (k * (k - 1))
        s += term
    return s


def cheb_fit(fun, a, b, n):
    """Coefficients c_0..c_{n-1} of sum' c_k T_k(t), t = (2x-a-b)/(b-a), from the n
    Chebyshev nodes (nodes, samples and the cosine sums all in 90-digit arithmetic)."""
    a, b = D(a), D(b)
    t = [dcos(PI * (D(j) + D("0.5")) / n) for j in range(n)]
    f = [fun((a + b) / 2 + (b - a) / 2 * tj) for tj in t]
    c = []
    for k in range(n):
        s = D(0)
        for j in range(n):
            s += f[j] * dcos(PI * k * (D(j) + D("0.5")) / n)
        c.append(s * 2 / n)
    c[0] /= 2
    return [float(v) for v in c]


def clenshaw(c, t):
    b1 = np.zeros_like(t)
    b2 = np.zeros_like(t)
    for ck in c[:0:-1]:
        b1, b2 = ck + 2 * t * b1 - b2, b1
    return c[0] + t * b1 - b2


def trim(c, tol=2e-18):
    n = len(c)
    while n > 2 and abs(c[n - 1]) < tol * abs(c[0]):
        n -= 1
    return c[:n]


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "bart_amd", "csrc", "expint_coef.hpp")
    small = trim(cheb_fit(q_small, 0.0, 1.0, 26))
    edges = [float(2.0 ** (j / 2.0)) for j in range(21)]
    large = [trim(cheb_fit(f_large, edges[j], edges[j + 1], 24)) for j in range(20)]
    nl = max(len(c) for c in large)
    # accuracy of the tables evaluated as on the device
    rng = np.random.default_rng(0)
    worst_s = worst_l = 0.0
    xs = np.concatenate([rng.uniform(0, 1, 300), 10.0 ** rng.uniform(-12, 0, 100)])
    for x in xs:
        got = x * np.log(x) + clenshaw(small, np.array(2 * x - 1.0))
        ref = e2(D(float(x)))
        worst_s = max(worst_s, abs(float((D(float(got)) - ref) / ref)))
    for j in range(20):
        a, b = edges[j], edges[j + 1]
        if a >= 709.0:
            continue
        for x in rng.uniform(a, min(b, 700.0), 40):   # e^-x is subnormal beyond
            t = (2 * x - (a + b)) / (b - a)
            got = np.exp(-x) * clenshaw(large[j], np.array(t))
            ref = e2(D(float(x)))
            worst_l = max(worst_l, abs(float((D(float(got)) - ref) / ref)))
    print("terms: small %d, large max %d; worst relative error: x<=1 %.2e, x>1 %.2e"
          % (len(small), nl, worst_s, worst_l))
    with open(out, "w") as f:
        f.write("// Generated by tools/gen_expint_tables.py -- do not edit.\n")
        f.write("// Chebyshev tables for E_2(x); see the generator for the definitions.\n")
        f.write("// Worst relative error of the fp64 evaluation against 90-digit references:\n")
        f.write("// %.1e on (0, 1], %.1e on (1, 700].\n#pragma once\n\nnamespace bartrt {\n\n" % (worst_s, worst_l))
        f.write("constexpr int kE2SmallTerms = %d;\n" % len(small))
        f.write("constexpr int kE2LargeTerms = %d;   // per half octave, zero padded\n" % nl)
        f.write("constexpr int kE2LargeIntervals = 20;  // [2^(j/2), 2^((j+1)/2)), j = 0..19\n\n")
        f.write("__device__ const double kE2Small[kE2SmallTerms] = {\n")
        f.write(",\n".join("    %.17e" % v for v in small) + "};\n\n")
        f.write("// per interval: t = x * scale + offset, then kE2LargeTerms coefficients\n")
        f.write("__device__ const double kE2Large[kE2LargeIntervals][2 + kE2LargeTerms] = {\n")
        rows = []
        for j in range(20):
            a, b = edges[j], edges[j + 1]
            c = large[j] + [0.0] * (nl - len(large[j]))
            rows.append("    {" + ", ".join("%.17e" % v for v in [2.0 / (b - a), -(a + b) / (b - a)] + c) + "}")
        f.write(",\n".join(rows) + "};\n\n}  // namespace bartrt\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
