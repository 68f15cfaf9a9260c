"""Line-by-line (Voigt) extinction oracle: numpy + scipy.special.wofz.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference's line-by-line code
lives in the absent transit submodule).  What is in the reference tree and is
followed here: the per-line fields of a TLI (doc/BART_user_manual/
BART_user_manual.tex:449-455), the Doppler and Lorentz half-widths
(scripts/broadening.py:121-127, 143) and the cfg keys nwidth / ethresh
(examples/demo/transit_demo.cfg:42-44, 57-58).  The Voigt function itself is
scipy's Faddeeva routine (exact to ~1e-13), deliberately a different algorithm
from the product's rational approximation.

Sampling (`wnosamp`, examples/demo/transit_demo.cfg:27-29; convention C15 of
DESIGN.md, SURVEY.md App. A-5 as recalled, unverified): the line sums of a layer
are evaluated on a grid `dv` times finer than the output grid and reduced to it.
dv is the smallest divisor of wnosamp whose spacing wndelt / dv is at most half
the narrowest line half-width of the layer (over the isotopes: max(Doppler HWHM
at the grid's low end, Lorentz HWHM)); rule "full" takes dv = wnosamp everywhere.
Reduction (Transit's `downsample` as recalled): an odd factor averages the dv
fine points centred on the output point; an even factor takes dv + 1 points with
the two end points at half weight, over dv; the first / last output point use
the half of that window that lies on the grid, normalised by its own weights.
wnosamp = 1: the line sums are evaluated on the output points themselves.

Voigt evaluation (cfg key `voigt` of the product; SURVEY.md App. A-5 as recalled,
unverified): "exact" evaluates the Faddeeva function per (line, point) at the line's own widths.
"grid" is the width-grid form: ndop x nlor half-widths, log spaced between dmin .. dmax and
lmin .. lmax (defaults: the Doppler half-widths of the TLI's isotopes over the kept line centres
and tlow .. thigh, the Lorentz half-widths over the atmosphere file's layers at those two
temperatures); a line takes the profile of the NEAREST grid widths (in the logarithm), centred on
the sampling point nearest to its centre, reaching floor(nwidth max(aD_grid, aL_grid) / step)
points either side; the profile values are Faddeeva values at the grid widths.
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


def divisors(n: int):
    return [d for d in range(1, n + 1) if n % d == 0]


def downsample(fine: np.ndarray, dv: int) -> np.ndarray:
    """fine[(W-1) dv + 1] -> out[W] (see the module docstring)."""
    if dv == 1:
        return fine.copy()
    W = (len(fine) - 1) // dv + 1
    h = dv // 2
    w = np.ones(2 * h + 1)
    if dv % 2 == 0:
        w[0] = w[-1] = 0.5
    out = np.zeros(W)
    for i in range(W):
        c = i * dv
        lo, hi = max(c - h, 0), min(c + h, len(fine) - 1)
        ww = w[lo - (c - h): hi - (c - h) + 1]
        out[i] = np.dot(ww, fine[lo:hi + 1]) / ww.sum()
    return out


class LblOracle:
    def __init__(self, tcfg: str, osamp_rule: str = "divisor", wn_slice=None, voigt=None):
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
        self.wn_ref = self.wn[0]              # Doppler widths for dv: at the FULL grid's low end
        self.wn_first, self.wn_last = self.wn[0], self.wn[-1]
        self.wndelt = d
        self.osamp = max(1, int(float(k.get("wnosamp", 1))))
        self.osamp_rule = osamp_rule
        full = self.wn
        if wn_slice is not None:              # a block of the grid (its edges keep the full grid's)
            self.wn = self.wn[wn_slice[0]:wn_slice[1]]
        self.voigt = voigt or k.get("voigt", "exact")
        assert self.voigt in ("exact", "grid")
        if self.voigt == "grid":
            self._width_grids(k, full, atm)

    def _width_grids(self, k, full, atm):
        """The Doppler and Lorentz half-width grids as Engine / lbl_init lays them out."""
        kept = np.concatenate([db["wn"][(db["wn"] >= full[0] - 500.0) & (db["wn"] <= full[-1] + 500.0)]
                               for db in self.dbs])
        nu_lo, nu_hi = (kept.min(), kept.max()) if len(kept) else (full[0], full[-1])
        tlo, thi = float(k.get("tlow", 500.0)), float(k.get("thigh", 3000.0))
        ih2 = self.species.index("H2") if "H2" in self.species else -1
        ihe = self.species.index("He") if "He" in self.species else -1
        dmin, dmax, lmin, lmax = np.inf, 0.0, np.inf, 0.0
        for db in self.dbs:
            sp = self.species.index(db["molecule"])
            for info in db["isotopes"]:
                mi = info["mass"] * AMU
                dop = lambda T: np.sqrt(2.0 * np.log(2.0) * KB * T / mi) / LS
                dmin, dmax = min(dmin, nu_lo * dop(tlo)), max(dmax, nu_hi * dop(thi))
                for l in range(len(self.press)):
                    s = sum(self.abund0[l][c] * (0.5 * (self.diam[sp] + self.diam[c])) ** 2
                            * np.sqrt(1.0 / mi + 1.0 / (self.mass[c] * AMU)) for c in (ih2, ihe) if c >= 0)
                    lor = lambda T: np.sqrt(2.0) / (LS * np.sqrt(np.pi * KB * T)) * self.press[l] * s
                    if s > 0:
                        lmin, lmax = min(lmin, lor(thi)), max(lmax, lor(tlo))
        if not lmax > 0:
            lmin = lmax = 1e-30
        dmin, dmax = float(k.get("dmin", dmin)), float(k.get("dmax", dmax))
        lmin, lmax = float(k.get("lmin", lmin)), float(k.get("lmax", lmax))
        nd, nl = int(float(k.get("ndop", 40))), int(float(k.get("nlor", 40)))

        def logspace(a, b, n):
            ln0 = np.log(a) if n > 1 else 0.5 * (np.log(a) + np.log(b))
            dln = (np.log(b) - np.log(a)) / (n - 1) if n > 1 else 0.0
            return np.exp(ln0 + dln * np.arange(n)), ln0, dln
        self.dgrid, self.d_ln0, self.d_dln = logspace(dmin, dmax, nd)
        self.lgrid, self.l_ln0, self.l_dln = logspace(lmin, lmax, nl)

    @staticmethod
    def _nearest(a, ln0, dln, n):
        if n <= 1 or not dln > 0:
            return np.zeros(np.shape(a), int)
        return np.clip(np.floor((np.log(a) - ln0) / dln + 0.5), 0, n - 1).astype(int)

    def layer_dv(self, T, p, q):
        """Oversampling factor of a layer (module docstring)."""
        if self.osamp == 1:
            return 1
        if self.osamp_rule == "full":
            return self.osamp
        ih2 = self.species.index("H2") if "H2" in self.species else -1
        ihe = self.species.index("He") if "He" in self.species else -1
        wmin = np.inf
        for db in self.dbs:
            sp = self.species.index(db["molecule"])
            for info in db["isotopes"]:
                mi = info["mass"] * AMU
                aD = self.wn_ref / LS * np.sqrt(2.0 * np.log(2.0) * KB * T / mi)
                s = 0.0
                for c in (ih2, ihe):
                    if c >= 0:
                        s += q[c] * (0.5 * (self.diam[sp] + self.diam[c])) ** 2 * \
                            np.sqrt(1.0 / mi + 1.0 / (self.mass[c] * AMU))
                aL = np.sqrt(2.0) / (LS * np.sqrt(np.pi * KB * T)) * p * s
                wmin = min(wmin, max(aD, aL))
        for dv in divisors(self.osamp):
            if self.wndelt / dv <= 0.5 * wmin:
                return dv
        return self.osamp

    def _layer(self, T, p, q, per_gram):
        """Extinction per database at one state, on the output grid: list of [W]
        arrays (evaluated dv times finer and reduced, see the module docstring)."""
        dv = self.layer_dv(T, p, q)
        if dv == 1 and self.voigt == "exact":
            return self._layer_on(self.wn, T, p, q, per_gram)
        # fine grid over this block plus half an output spacing either side, clipped to the full grid
        h = dv // 2
        k0 = int(round((self.wn[0] - self.wn_first) / self.wndelt)) * dv
        k1 = int(round((self.wn[-1] - self.wn_first) / self.wndelt)) * dv
        kmax = int(round((self.wn_last - self.wn_first) / self.wndelt)) * dv
        ka, kb = max(k0 - h, 0), min(k1 + h, kmax)
        fine_x = self.wn_first + np.arange(ka, kb + 1) * (self.wndelt / dv)
        w = np.ones(2 * h + 1)
        if dv % 2 == 0:
            w[0] = w[-1] = 0.5
        out = []
        fines = (self._layer_on(fine_x, T, p, q, per_gram) if self.voigt == "exact"
                 else self._layer_grid(ka, kb, dv, T, p, q, per_gram))
        for fine in fines:
            if dv == 1:
                out.append(np.array(fine[k0 - ka: k1 - ka + 1]))
                continue
            o = np.zeros(len(self.wn))
            for i in range(len(self.wn)):
                c = k0 + i * dv
                lo, hi = max(c - h, 0), min(c + h, kmax)
                ww = w[lo - (c - h): hi - (c - h) + 1]
                o[i] = np.dot(ww, fine[lo - ka: hi - ka + 1]) / ww.sum()
            out.append(o)
        return out

    def _strengths(self, db, T, p, q, per_gram):
        """Per line of a database at one state: strength, Doppler and Lorentz half-widths."""
        ih2 = self.species.index("H2") if "H2" in self.species else -1
        ihe = self.species.index("He") if "He" in self.species else -1
        sp = self.species.index(db["molecule"])
        nu0, iso = db["wn"], db["iso"]
        S = np.zeros(len(nu0)); aD = np.zeros(len(nu0)); aL = np.zeros(len(nu0))
        for i, info in enumerate(db["isotopes"]):
            m = iso == i
            mi = info["mass"] * AMU
            Z = np.interp(T, db["temps"], info["Z"])
            scale = info["ratio"] / (Z * self.mass[sp] * AMU) if per_gram else info["ratio"] * q[sp] * p / (KB * T) / Z
            S[m] = (SIGCTE * scale * db["gf"][m] * np.exp(-EXPCTE * db["elow"][m] / T)
                    * (1.0 - np.exp(-EXPCTE * nu0[m] / T)))
            aD[m] = nu0[m] / LS * np.sqrt(2.0 * np.log(2.0) * KB * T / mi)
            s = sum(q[c] * (0.5 * (self.diam[sp] + self.diam[c])) ** 2 * np.sqrt(1.0 / mi + 1.0 / (self.mass[c] * AMU))
                    for c in (ih2, ihe) if c >= 0)
            aL[m] = np.sqrt(2.0) / (LS * np.sqrt(np.pi * KB * T)) * p * s
        return S, aD, aL

    def _layer_grid(self, ka, kb, dv, T, p, q, per_gram):
        """Width-grid form (module docstring) on the sampling points ka .. kb of the full grid's
        dv-fold refinement: list of arrays, one per database."""
        step, inv_step = self.wndelt / dv, dv / self.wndelt
        sl2 = np.sqrt(np.log(2.0))
        out, cache = [], {}
        for db in self.dbs:
            e = np.zeros(kb - ka + 1)
            inr = (db["wn"] >= self.wn_first - 500.0) & (db["wn"] <= self.wn_last + 500.0)   # the lines the engine keeps
            S, aD, aL = self._strengths(db, T, p, q, per_gram)
            S = np.where(inr, S, 0.0)
            keep = (S >= self.ethresh * S.max()) & (S > 0)
            iD = self._nearest(np.where(keep, aD, 1.0), self.d_ln0, self.d_dln, len(self.dgrid))
            iL = self._nearest(np.where(keep, np.maximum(aL, 1e-300), 1.0), self.l_ln0, self.l_dln, len(self.lgrid))
            for j in np.where(keep)[0]:
                gd, gl = self.dgrid[iD[j]], self.lgrid[iL[j]]
                K = int(np.floor(self.nwidth * max(gd, gl) / step))
                kc = int(np.floor((db["wn"][j] - self.wn_first) * inv_step + 0.5))
                lo, hi = max(kc - K, ka), min(kc + K, kb)
                if lo > hi:
                    continue
                key = (iD[j], iL[j])
                if key not in cache:
                    o = np.arange(K + 1)
                    cache[key] = sl2 / np.sqrt(np.pi) / gd * wofz(sl2 * (o * step + 1j * gl) / gd).real
                prof = cache[key]
                k = np.arange(lo, hi + 1)
                e[lo - ka: hi - ka + 1] += S[j] * prof[np.abs(k - kc)]
            out.append(e)
        return out

    def _layer_on(self, grid, T, p, q, per_gram):
        """Extinction per database at one state on `grid`: list of arrays."""
        out = []
        ih2 = self.species.index("H2") if "H2" in self.species else -1
        ihe = self.species.index("He") if "He" in self.species else -1
        for db in self.dbs:
            sp = self.species.index(db["molecule"])
            e = np.zeros(len(grid))
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
            near = keep & (nu0 + cut >= grid[0]) & (nu0 - cut <= grid[-1])
            for j in np.where(near)[0]:
                a, b = np.searchsorted(grid, [nu0[j] - cut[j], nu0[j] + cut[j]])
                w = np.arange(max(a - 1, 0), min(b + 1, len(grid)))
                w = w[np.abs(grid[w] - nu0[j]) <= cut[j]]
                if len(w) == 0:
                    continue
                x = sl2 * np.abs(grid[w] - nu0[j]) / aD[j]
                y = sl2 * aL[j] / aD[j]
                e[w] += S[j] * sl2 / np.sqrt(np.pi) / aD[j] * wofz(x + 1j * y).real
            out.append(e)
        return out

    def extinction(self, prof, layers=None):
        """prof [(S+1), L] -> ext [L, W] in cm-1 (atm layer order); `layers`: only
        these rows are evaluated (the others stay 0)."""
        prof = np.asarray(prof, float).reshape(len(self.species) + 1, -1)
        L = prof.shape[1]
        ext = np.zeros((L, len(self.wn)))
        for l in (range(L) if layers is None else layers):
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
