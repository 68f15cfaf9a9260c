"""One-time host-side readers of the per-step callable's inputs.

Counterparts (same inputs, same outputs, independent implementation) of the
reference helpers the worker calls once at start-up:

* ``TepFile``          -- reference code/reader.py:67-137 (``File.getvalue``)
* ``readatm``          -- reference code/makeatm.py:758-837
* ``readfilter``       -- reference code/wine.py:16-66
* ``read_kurucz``      -- reference code/kurucz_inten.py:162-317
* ``readkurucz``       -- reference code/wine.py:69-124
* ``resample``         -- reference code/wine.py:127-174
* ``bandintegrate``    -- reference code/wine.py:177-199 (host check only; the
                          per-step integration runs on the GPU)
"""
from __future__ import annotations

import numpy as np

# code/constants.py:7-19
Mjup, Rjup, Rsun = 1.8983e+27, 7.1492e+07, 6.96e+08
H, LS, KB = 6.6260755e-27, 2.99792458e10, 1.380658e-16
sig = 5.670367e-8
# scipy.constants values used by the reference worker (CODATA 2018)
AU, G_NEWTON, C_LIGHT = 149597870700.0, 6.67430e-11, 299792458.0


def _trapz(y, x):
    y, x = np.asarray(y, float), np.asarray(x, float)
    return float(np.sum(np.diff(x) * (y[1:] + y[:-1]) / 2.0))


class TepFile:
    """``parameter value ...`` lines, ``#`` comments; values kept as strings and
    converted to numbers where they parse (reader.py:67-129)."""

    def __init__(self, path: str):
        self.entries = {}
        for line in open(path):
            line = line.split("#", 1)[0].strip()
            if line:
                tok = line.split()
                self.entries.setdefault(tok[0], tok[1:])

    def getvalue(self, par: str):
        """List of the entry's values, numbers where they parse (NaN if absent);
        callers index [0] exactly as with the reference reader."""
        if par not in self.entries:
            return np.nan
        out = []
        for v in self.entries[par]:
            for conv in (int, float):
                try:
                    out.append(conv(v))
                    break
                except ValueError:
                    continue
            else:
                out.append(v)
        return out

    def num(self, par: str) -> float:
        return float(self.entries[par][0])


def readatm(path: str):
    """-> (species list, pressure[bar], temperature[K], abundances[L, S]); the
    radius column is skipped when present (makeatm.py:810-835)."""
    lines = open(path).read().split("\n")
    isp = lines.index("#SPECIES")
    species = lines[isp + 1].split()
    start = lines.index("#TEADATA") + 2
    rows = [ln.split() for ln in lines[start:] if ln.strip()]
    d = np.array(rows, dtype=np.double)
    off = 0 if d.shape[1] == len(species) + 2 else 1
    return species, d[:, off].copy(), d[:, off + 1].copy(), d[:, off + 2:].copy()


def readfilter(path: str):
    """Two columns (wavelength um, response) -> (wavenumber cm-1 ascending,
    response), i.e. both reversed (wine.py:58-66)."""
    rows = []
    started = False
    for line in open(path):
        s = line.strip()
        if not started and (not s or s.startswith("#")):
            continue
        started = True
        if s:
            rows.append(s.split()[:2])
    d = np.array(rows, dtype=np.double)[::-1]
    return 1.0 / (d[:, 0] * 1e-4), d[:, 1].copy()


def read_kurucz(path: str, freq: bool = False):
    """Kurucz ``.pck`` grid: 10-character fields; ``TEFF`` header per model with
    T in columns 5:12 and log g in 22:29; wavelength block (nm) after the line
    ending in ``END``; two blocks per model (with / without lines).
    -> (inten, wave, grav, temp, nainten, head) as kurucz_inten.read returns."""
    txt = open(path).read().replace("\r", "\n").split("\n")
    heads = [i for i, ln in enumerate(txt) if ln.startswith("TEFF")]
    # the reference drops a model whose header sits on line 0 and any T == 0;
    # real files never do either
    temp = np.array([float(txt[i][5:12]) for i in heads])
    grav = np.array([float(txt[i][22:29]) for i in heads])
    head = [txt[i] for i in heads]
    startwave = max(i for i, ln in enumerate(txt) if ln.endswith("END")) + 1
    nline = (heads[2] - heads[1] - 1) // 2 if len(heads) > 2 else (len(txt) - heads[0] - 1) // 2

    def fields(block):
        s = "".join(block)
        return np.array([float(s[j:j + 10]) for j in range(0, len(s), 10) if s[j:j + 10].strip()])

    wave = fields(txt[startwave:heads[0]])
    wave = wave[wave != 0] * 1e-9
    n = wave.size
    inten = np.zeros((len(heads), n))
    nain = np.zeros((len(heads), n))
    for m, h in enumerate(heads):
        a = fields(txt[h + 1:h + 1 + nline])
        b = fields(txt[h + 1 + nline:h + 1 + 2 * nline])
        inten[m, :min(n, a.size)] = a[:n]
        nain[m, :min(n, b.size)] = b[:n]
    inten *= 4.0 * 1e-3
    nain *= 4.0 * 1e-3
    if freq:
        wave = (C_LIGHT / wave)[::-1].copy()
        inten = inten[:, ::-1].copy()
        nain = nain[:, ::-1].copy()
    return inten, wave, grav, temp, nain, head


def readkurucz(path: str, temperature: float, logg: float):
    """Nearest-temperature, nearest-or-higher log g model -> stellar flux per
    wavenumber, erg s-1 cm-2 cm (wine.py:99-124)."""
    inten, freq, grav, temp, _, _ = read_kurucz(path, freq=True)
    starwn = freq / C_LIGHT * 1e-2
    tmodel = temp[np.argmin(np.abs(temp - temperature))]
    gmodel = grav[np.argmin(np.abs(grav - logg))]
    imodel = np.where((temp == tmodel) & (grav >= gmodel))[0][0]
    starfl = inten[imodel] * 1e3 * np.pi * (1e2 * C_LIGHT)
    return starfl, starwn, tmodel, gmodel


def _interp_lin(x, xp, fp):
    """scipy.interpolate.interp1d(kind='linear') on an in-range x: slope form
    anchored at the lower node, as scipy evaluates it."""
    x = np.asarray(x, float)
    hi = np.clip(np.searchsorted(xp, x), 1, len(xp) - 1)
    lo = hi - 1
    slope = (fp[hi] - fp[lo]) / (xp[hi] - xp[lo])
    return slope * (x - xp[lo]) + fp[lo]


def resample(specwn, filterwn, filtertr, starwn, starfl):
    """Filter and star on the spectrum samples strictly inside the filter's
    range; filter normalised to unit integral (wine.py:158-174).
    -> (nifilter, istarfl, wnindices) with wnindices a tuple like np.where's."""
    specwn = np.asarray(specwn, float)
    idx = np.where((specwn < filterwn[-1]) & (filterwn[0] < specwn))
    x = specwn[idx]
    istar = _interp_lin(x, np.asarray(starwn, float), np.asarray(starfl, float))
    ifil = _interp_lin(x, np.asarray(filterwn, float), np.asarray(filtertr, float))
    return ifil / _trapz(ifil, x), istar, idx


def bandintegrate(spectrum, specwn, nifilter, wnindices):
    return _trapz(np.asarray(spectrum) * nifilter, np.asarray(specwn)[wnindices])
