"""Line-by-line (Voigt) extinction oracle: numpy + scipy.special.wofz.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference's line-by-line code
lives in the absent transit submodule).  What is in the reference tree and is
followed here: the per-line fields of a TLI (doc/BART_user_manual/
BART_user_manual.tex:449-455), the Doppler and Lorentz half-widths
(scripts/broadening.py:121-127, 143) and the cfg keys nwidth / ethresh
(examples/demo/transit_demo.cfg:42-44, 57-58).  The Voigt function itself is
scipy's Faddeeva routine (exact to ~1e-13), deliberately a different algorithm
from the product's rational approximation.
"""
from __future__ import annotations

import struct

import numpy as np
from scipy.special import wofz

from .rt_oracle import AMU, H, KB, LS, read_atm, read_molfile, read_tcfg

SIGCTE = 8.852821681767784e-13   # pi e^2 / (m_e c^2), cm
EXPCTE = H * LS / KB


def read_tli(path: str):
    b = open(path, "rb").read()
    off = 0

    def get(fmt):
        nonlocal off
        v = struct.unpack_from("=" + fmt, b, off)
        off += struct.calcsize("=" + fmt)
        return v if len(v) > 1 else v[0]

    def gstr():
        nonlocal off
        n = get("H")
        s = b[off:off + n].decode()
        off += n
        return s

    def arr(dt, n):
        nonlocal off
        a = np.frombuffer(b, dt, n, off).copy()
        off += a.nbytes
        return a

    assert get("i") == 0x494C54FF
    get("3H")
    wn_lo, wn_hi = get("2d")
    dbs = []
    for _ in range(get("H")):
        db = {"name": gstr(), "molecule": gstr()}
        nt, ni = get("HH")
        db["temps"] = arr(np.float64, nt)
        db["isotopes"] = []
        for _ in range(ni):
            name = gstr()
            mass, ratio = get("dd")
            db["isotopes"].append({"name": name, "mass": mass, "ratio": ratio, "Z": arr(np.float64, nt)})
        dbs.append(db)
    for db in dbs:
        n = get("q")
        db["wn"] = arr(np.float64, n)
        db["iso"] = arr(np.int16, n)
        db["elow"] = arr(np.float64, n)
        db["gf"] = arr(np.float64, n)
    return dbs


class LblOracle:
    def __init__(self, tcfg: str):
        k = read_tcfg(tcfg)
        self.keys = k
        atm = read_atm(k["atm"])
        mol = read_molfile(k["molfile"])
        self.species = atm["species"]
        self.press = atm["press"]
        self.abund0 = atm["abund"]
        self.mass = np.array([mol["mass"][mol["name"].index(s)] for s in self.species])
        self.diam = np.array([mol["diam"][mol["name"].index(s)] for s in self.species]) * 1e-8
        self.molid = {n: i for n, i in zip(mol["name"], mol["id"])}
        self.dbs = [db for f in k["linedb"].replace(",", " ").split() for db in read_tli(f)]
        self.nwidth = float(k.get("nwidth", 20))
        self.ethresh = float(k.get("ethresh", 1e-6))
        lo, hi, d = float(k["wnlow"]), float(k["wnhigh"]), float(k.get("wndelt", 1.0))
        self.wn = lo + d * np.arange(int(np.floor((hi - lo) / d + 1e-9)) + 1)

    def _layer(self, T, p, q, per_gram):
        """Extinction per database at one state: list of [W] arrays."""
        out = []
        ih2 = self.species.index("H2") if "H2" in self.species else -1
        ihe = self.species.index("He") if "He" in self.species else -1
        for db in self.dbs:
            sp = self.species.index(db["molecule"])
            e = np.zeros(len(self.wn))
            nu0, iso = db["wn"], db["iso"]
            S = np.zeros(len(nu0)); aD = np.zeros(len(nu0)); aL = np.zeros(len(nu0))
            for i, info in enumerate(db["isotopes"]):
                m = iso == i
                mi = info["mass"] * AMU
                Z = np.interp(T, db["temps"], info["Z"])
                if per_gram:
                    scale = info["ratio"] / (Z * self.mass[sp] * AMU)
                else:
                    scale = info["ratio"] * q[sp] * p / (KB * T) / Z
                S[m] = (SIGCTE * scale * db["gf"][m] * np.exp(-EXPCTE * db["elow"][m] / T)
                        * (1.0 - np.exp(-EXPCTE * nu0[m] / T)))
                aD[m] = nu0[m] / LS * np.sqrt(2.0 * np.log(2.0) * KB * T / mi)
                s = 0.0
                for c in (ih2, ihe):
                    if c >= 0:
                        s += q[c] * (0.5 * (self.diam[sp] + self.diam[c])) ** 2 * \
                            np.sqrt(1.0 / mi + 1.0 / (self.mass[c] * AMU))
                aL[m] = np.sqrt(2.0) / (LS * np.sqrt(np.pi * KB * T)) * p * s
            keep = (S >= self.ethresh * S.max()) & (S > 0)
            cut = self.nwidth * np.maximum(aD, aL)
            sl2 = np.sqrt(np.log(2.0))
            for j in np.where(keep)[0]:
                w = np.where(np.abs(self.wn - nu0[j]) <= cut[j])[0]
                if len(w) == 0:
                    continue
                x = sl2 * np.abs(self.wn[w] - nu0[j]) / aD[j]
                y = sl2 * aL[j] / aD[j]
                e[w] += S[j] * sl2 / np.sqrt(np.pi) / aD[j] * wofz(x + 1j * y).real
            out.append(e)
        return out

    def extinction(self, prof):
        """prof [(S+1), L] -> ext [L, W] in cm-1 (atm layer order)."""
        prof = np.asarray(prof, float).reshape(len(self.species) + 1, -1)
        L = prof.shape[1]
        ext = np.zeros((L, len(self.wn)))
        for l in range(L):
            ext[l] = sum(self._layer(prof[0, l], self.press[l], prof[1:, l], False))
        return ext

    def opacity_table(self, tgrid):
        """o[L][Nt][M][W] in cm2/g, molecules in database order."""
        L = len(self.press)
        o = np.zeros((L, len(tgrid), len(self.dbs), len(self.wn)))
        for l in range(L):
            for t, T in enumerate(tgrid):
                o[l, t] = np.array(self._layer(T, self.press[l], self.abund0[l], True))
        return o
