"""Coefficients of bart_amd/csrc/voigt_coef.hpp: Weideman (1994, SIAM J. Numer. Anal.
31, 1497), w(z) = 2 p(Z)/(L - iz)^2 + pi^-1/2/(L - iz), Z = (L + iz)/(L - iz), p of degree
N - 1 from the paper's FFT recipe, L = sqrt(N / sqrt 2).

    python tools/gen_voigt_coef.py [N]     prints the table (highest degree first) and its
                                           error against scipy's wofz over |z| < 8
    python tools/gen_voigt_coef.py --check compares the recipe with the committed header

Accuracy over |z| < 8, y >= 1e-9 (absolute against the line-centre value 1 / relative where
the function is above 1e-6): N = 32: 4e-14 / 4e-8, 36: 1.3e-15 / 8e-10, 40: 1.1e-15 / 1e-10."""
import os
import re
import sys

import numpy as np


def coef(N):
    M = 2 * N
    k = np.arange(-M + 1, M)
    L = np.sqrt(N / np.sqrt(2.0))
    t = L * np.tan(k * np.pi / M / 2)
    f = np.concatenate([[0.0], np.exp(-t * t) * (L * L + t * t)])
    a = np.real(np.fft.fft(np.fft.fftshift(f))) / (2 * M)
    return a[1:N + 1][::-1], L


def evaluate(a, L, x, y):
    z = x + 1j * y
    Z = (L + 1j * z) / (L - 1j * z)
    return (2 * np.polyval(a, Z) / (L - 1j * z) ** 2 + (1 / np.sqrt(np.pi)) / (L - 1j * z)).real


if __name__ == "__main__":
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bart_amd", "csrc", "voigt_coef.hpp")
    if "--check" in sys.argv:
        src = open(hdr).read()
        tab = np.array([float(v) for v in re.findall(r"-?\d\.\d+e[+-]\d+", src.split("kWeidA")[1])])
        a, L = coef(len(tab))
        print("N =", len(tab), "max |recipe - header| =", np.abs(a - tab).max())
        sys.exit(0 if np.array_equal(a, tab) else 1)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 36
    a, L = coef(N)
    from scipy.special import wofz
    rng = np.random.default_rng(3)
    x, y = rng.uniform(0, 8, 600000), 10 ** rng.uniform(-9, 0.9, 600000)
    m = x * x + y * y < 64
    x, y = x[m], y[m]
    ref, v = wofz(x + 1j * y).real, evaluate(a, L, x[...], y[...])
    big = ref > 1e-6
    print("// N = %d, L = %.17g; abs err %.2g, rel err (function > 1e-6) %.2g" %
          (N, L, np.abs(v - ref).max(), np.abs(v[big] / ref[big] - 1).max()))
    for i in range(0, N, 3):
        print("    " + " ".join("%.17e," % c for c in a[i:i + 3]))
