"""Synthetic inputs of the per-step callable that sit outside the engine: a TEP
file, filter curves, a stellar grid and the ``[MCMC]`` configuration the worker
reads (reference examples/demo/BART_eclipse.cfg).  Stand-ins for files that are
not shipped to the GPU box."""
from __future__ import annotations

import os

import numpy as np

from . import synth

# values of inputs/tep/HD209458b.tep (Ts, Rs, a, Rp, Mp, loggstar)
HD209458B = {"Ts": 6075, "Rs": 1.145, "a": 0.047, "Rp": 1.350, "Mp": 0.66, "loggstar": 4.37}


def write_tep(path: str, values=None) -> None:
    v = dict(HD209458B if values is None else values)
    with open(path, "w") as f:
        f.write("# synthetic TEP file: parameter value uncert unit\n")
        f.write("planetname      synth     -1        -\n")
        for k, x in v.items():
            f.write("%-15s %-13s -1        -   # comment\n" % (k, repr(x)))


def write_filters(outdir: str, wl_lo_um: float, wl_hi_um: float, n: int = 10, npts: int = 60):
    """n smooth band-passes tiling (wl_lo, wl_hi) microns, two-column files."""
    os.makedirs(outdir, exist_ok=True)
    edges = np.linspace(wl_lo_um, wl_hi_um, n + 1)
    files = []
    for i in range(n):
        lo, hi = edges[i] + 0.002, edges[i + 1] - 0.002
        wl = np.linspace(lo, hi, npts)
        x = (wl - lo) / (hi - lo)
        tr = np.clip(np.sin(np.pi * x) ** 0.5 * (0.8 + 0.2 * x), 0.0, None)
        p = os.path.join(outdir, "fsynth%02d.dat" % (i + 1))
        synth.write_filter(p, wl, tr)
        files.append(p)
    return files


def make_worker_case(outdir: str, nwave=2501, wnlow=2500.0, opmol=("CH4",), molfit=("CH4",),
                     params=(-2.0, 0.0, 1.0, 0.0, 0.98, -0.5), nfilters=10, solution="eclipse",
                     ebalance=False, tep=None, **case_kw):
    """Engine inputs (synth.make_case) + TEP + filters + star + MCMC cfg.
    Defaults mirror examples/demo/BART_eclipse.cfg (CH4, 2-4 um, 10 filters);
    `tep`: the TEP file's values (default HD209458B above)."""
    case = synth.make_case(outdir, nwave=nwave, wnlow=wnlow, opmol=opmol, **case_kw)
    tep_values = tep
    tep = os.path.join(outdir, "planet.tep")
    write_tep(tep, tep_values)
    star = os.path.join(outdir, "star.pck")
    synth.blackbody_kurucz(star)
    wl_hi = 1e4 / (case.wn[0] + 2.0)
    wl_lo = 1e4 / (case.wn[-1] - 2.0)
    filters = write_filters(os.path.join(outdir, "filters"), wl_lo, wl_hi, nfilters)
    cfg = os.path.join(outdir, "BART.cfg")
    with open(cfg, "w") as f:
        f.write("[MCMC]\n")
        f.write("tep_name = %s\nkurucz = %s\natmfile = %s\ntconfig = %s\n"
                % (tep, star, case.atm, case.tcfg))
        f.write("filters = " + "\n          ".join(filters) + "\n")
        f.write("func = hack BARTfunc ./\n")
        f.write("molfit = %s\n" % " ".join(molfit))
        f.write("Tmin = 400.0\nTmax = 3000.0\nPTtype = line\n")
        f.write("params = " + "  ".join(repr(float(p)) for p in params) + "\n")
        f.write("solution = %s\n" % solution)
        if ebalance:
            f.write("ebalance = True\n")
    return case, cfg
