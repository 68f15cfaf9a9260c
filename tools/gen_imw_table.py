"""Table of bart_amd/csrc/imw_tab.hpp: D(x) = Im w(x) = 2 F(x) / sqrt(pi) (F: Dawson's integral) on the real
axis: 0 <= x < 8 as 128 pieces of width 1/16 (the line cores; NW further pieces of width WH from WX0 can be
appended), each a polynomial of degree 7 in t = (x - x0) / h - i - 1/2 (|t| <= 1/2): the
interpolant through the piece's eight Chebyshev nodes, computed with 40 digits and rounded to double.  Coefficients 0-3 are kept as doubles, 4-7 as floats.  Relative error of the rounded polynomial on 40 points
per piece: printed into the header (the first piece, where D -> 0 and exp(-x^2) carries the function, apart).  The small-y branch of the kernels' Voigt function (csrc/lbl.hip, voigt_taylor) expands
w(x + i y) about the real axis from w(x) = exp(-x^2) + i D(x).

    python tools/gen_imw_table.py            prints the header
    python tools/gen_imw_table.py --check    compares with the committed header"""
import os
import re
import sys

import mpmath as mp

mp.mp.dps = 40
NP, DEG, XMAX = 128, 7, 8          # the core pieces
WX0, WH, NW = 8.0, 0.25, 0          # optional coarser pieces beyond x = 8 (none: see lbl.hip, voigt_taylor)


def imw(x):
    x = mp.mpf(x)
    return mp.exp(-x * x) * mp.erfi(x)          # = 2 F(x) / sqrt(pi)


def xof(i, t):
    """Abscissa of local coordinate t in row i of the table (core pieces, then wing pieces)."""
    if i < NP:
        return (i + mp.mpf(1) / 2 + t) * XMAX / NP
    return mp.mpf(WX0) + (i - NP + mp.mpf(1) / 2 + t) * mp.mpf(WH)


def piece(i):
    """Monomial coefficients in t of the Chebyshev interpolant of row i."""
    n = DEG + 1
    nodes = [mp.cos((k + mp.mpf(1) / 2) * mp.pi / n) / 2 for k in range(n)]          # t in [-1/2, 1/2]
    vals = [imw(xof(i, t)) for t in nodes]
    # solve the Vandermonde system exactly enough (40 digits, 8 x 8)
    A = mp.matrix(n, n)
    for r, t in enumerate(nodes):
        for c in range(n):
            A[r, c] = t ** c
    return list(mp.lu_solve(A, mp.matrix(vals)))


def table():
    """Coefficients 0-3 as doubles, 4-7 rounded to float (they multiply t^4 .. t^7, |t| <= 1/2: their
    rounding error stays below 1e-13 of D) -- a row is 48 bytes, and rows 48 bytes apart start on banks
    0, 12, 24, 4, 16, 28, 8, 20 of the 32 LDS banks: neighbouring pieces never share a bank."""
    import numpy as np
    rows = []
    for i in range(NP + NW):
        c = [float(v) for v in piece(i)]
        rows.append(c[:4] + [float(np.float32(v)) for v in c[4:]])
    return rows


def worst_error(tab, first=False):
    w = mp.mpf(0)
    for i, c in enumerate(tab):
        if (i == 0) != first:
            continue
        for k in range(40):
            t = mp.mpf(k) / 39 - mp.mpf(1) / 2
            ref = imw(xof(i, t))
            v = sum(mp.mpf(c[j]) * t ** j for j in range(DEG + 1))
            if ref != 0:
                w = max(w, abs(v / ref - 1))
    return float(w)


ERR = (0.0, 0.0)


def header(tab):
    out = ["// D(x) = Im w(x) = 2 F(x) / sqrt(pi): %d pieces of width 1/16 on 0 <= x < 8 (+ %d of width %g from" % (NP, NW, WH),
           "// x = %g); degree-%d polynomials in t = (x - x0) / h - i - 1/2, lowest degree first;" % (WX0, DEG),
           "// coefficients 0-3 double, 4-7 float (48-byte rows: neighbouring",
           "// pieces start on different LDS banks).  tools/gen_imw_table.py; relative error <= %.1e (%.1e in the" % ERR,
           "// first piece, where D -> 0).  Copied into LDS by the kernels that evaluate the Voigt function",
           "// (lbl.hip, voigt_taylor).",
           "#pragma once", "namespace bartrt {",
           "constexpr int kImwPieces = %d, kImwWing = %d, kImwRows = kImwPieces + kImwWing;" % (NP, NW),
           "constexpr double kImwWingX0 = %r, kImwWingInvH = %r;" % (WX0, 1.0 / WH),
           "struct ImwRow { double c[4]; float f[4]; };",
           "__device__ const ImwRow kImwTab[kImwRows] = {"]
    for c in tab:
        out.append("    {{" + ", ".join("%.17e" % v for v in c[:4]) + "}, {" + ", ".join("%.9ef" % v for v in c[4:]) + "}},")
    out += ["};", "}  // namespace bartrt"]
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bart_amd", "csrc", "imw_tab.hpp")
    tab = table()
    if "--check" in sys.argv:
        src = open(hdr).read().split("kImwTab[kImwRows] = {")[1]
        have = [float(v) for v in re.findall(r"-?\d\.\d+e[+-]\d+", src)]
        import numpy as np
        want = [v for c in tab for v in c]
        # the float columns are printed with nine digits: compare them as floats
        isf = [k % 8 >= 4 for k in range(len(want))]
        same = len(have) == len(want) and all((np.float32(a) == np.float32(b)) if f else (a == b)
                                              for a, b, f in zip(have, want, isf))
        print("rows %d, coefficients %d, identical to the header: %s" % (NP + NW, len(want), same))
        sys.exit(0 if same else 1)
    ERR = (worst_error(tab), worst_error(tab, first=True))
    sys.stderr.write("worst relative error over 40 points per piece: %.2e (first piece %.2e)\n" % ERR)
    sys.stdout.write(header(tab))
