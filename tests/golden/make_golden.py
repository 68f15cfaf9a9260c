"""Generates the golden vectors in this directory by IMPORTING the reference's
own Python modules from /root/reference/code (read-only; nothing is copied).
Run once in the build container:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Only inputs and expected outputs are stored (small .npz/.json).  The reference
cannot travel to the GPU box; these fixtures can.  numpy >= 1.24 removed
np.float/np.int, which a few reference lines off the per-step path still use
(SURVEY.md 8c), so they are aliased before the import.
"""
import json
import os
import sys
import warnings

import numpy as np

warnings.simplefilter("ignore")
np.float = float  # noqa
np.int = int      # noqa
REF = "/root/reference"
sys.dont_write_bytecode = True   # nothing is written into the (read-only) reference tree
sys.path.insert(0, os.path.join(REF, "code"))
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import PT as pt            # noqa: E402
import wine as w           # noqa: E402
import reader as rd        # noqa: E402
import makeatm as mat      # noqa: E402
import cf                  # noqa: E402
import constants as c      # noqa: E402
import kurucz_inten as ki  # noqa: E402
import scipy.constants as sc  # noqa: E402

from bart_amd import synth  # noqa: E402

rng = np.random.default_rng(20260102)
p = np.logspace(-5, 2, 100)          # bar, top -> bottom as PT_generator receives it

# ---------------------------------------------------------------- TEP + PTargs
tepf = os.path.join(REF, "inputs/tep/HD209458b.tep")
tep = rd.File(tepf)
tstar = float(tep.getvalue('Ts')[0])
rstar = float(tep.getvalue('Rs')[0]) * c.Rsun
sma = float(tep.getvalue('a')[0]) * sc.au
rplanet = float(tep.getvalue('Rp')[0]) * c.Rjup
mplanet = float(tep.getvalue('Mp')[0]) * c.Mjup
gstar = float(tep.getvalue('loggstar')[0])
gplanet = 100.0 * sc.G * mplanet / rplanet ** 2
misc = {"tep": {"Ts": tstar, "Rs_m": rstar, "a_m": sma, "Rp_m": rplanet, "Mp_kg": mplanet,
                "loggstar": gstar, "gplanet_cgs": gplanet, "rprs": rplanet / rstar}}
tep2 = rd.File(os.path.join(REF, "examples/WASP-12b/WASP-12b.tep"))
misc["tep_wasp12b"] = {k: float(tep2.getvalue(k)[0]) for k in ("Ts", "Rs", "a", "Rp", "Mp", "loggstar")}

# ---------------------------------------------------------------- T(p) models
N = 32
out = {"p": p}
# line (PT.py:589-701); parameter box of examples/demo/BART_eclipse.cfg:82-83
lo = np.array([-5.0, -2.0, -2.0, 0.0, 0.55]); hi = np.array([-1.0, 1.0, 1.0, 1.0, 1.2])
par = lo + (hi - lo) * rng.random((N, 5))
par[0] = [-2.0, 0.0, 1.0, 0.0, 0.98]      # the demo's starting point
out["line_params"] = par
out["line_args"] = np.array([rstar, tstar, 100.0, sma, gplanet])
out["line_T"] = np.array([pt.PT_generator(p, q, pt.PT_line, [rstar, tstar, 100.0, sma, gplanet, 'const'])
                          for q in par])
out["line_T_thorngren"] = np.array([pt.PT_generator(p, q, pt.PT_line,
                                                    [rstar, tstar, 100.0, sma, gplanet, 'thorngren'])
                                    for q in par[:8]])
out["xi_gamma"] = 10 ** rng.uniform(-2, 1, 64)
out["xi_tau"] = 10 ** rng.uniform(-6, 4, 64)
out["xi"] = np.array([pt.xi(g, np.array([t]))[0] for g, t in zip(out["xi_gamma"], out["xi_tau"])])
# iso
out["iso_params"] = rng.uniform(500, 2500, (4, 1))
out["iso_T"] = np.array([pt.PT_generator(p, q, pt.PT_iso) for q in out["iso_params"]])
# madhu_noinv: a1 a2 p1 p3 T3 ; madhu_inv: a1 a2 p1 p2 p3 T3 (docstring ranges PT.py:257-262,479-483)
pn = np.column_stack([rng.uniform(0.2, 0.6, N), rng.uniform(0.04, 0.5, N), rng.uniform(0.001, 0.01, N),
                      rng.uniform(0.5, 10, N), rng.uniform(1500, 1700, N)])
tn, okn = [], []
for q in pn:
    try:
        tn.append(pt.PT_generator(p, q, pt.PT_NoInversion)); okn.append(1)
    except ValueError:
        tn.append(np.zeros(100)); okn.append(0)
out["noinv_params"], out["noinv_T"], out["noinv_ok"] = pn, np.array(tn), np.array(okn)
pi_ = np.column_stack([rng.uniform(0.2, 0.6, N), rng.uniform(0.04, 0.5, N), rng.uniform(0.001, 0.01, N),
                       rng.uniform(0.01, 1, N), rng.uniform(0.5, 10, N), rng.uniform(1500, 1700, N)])
ti, oki = [], []
for q in pi_:
    try:
        ti.append(pt.PT_generator(p, q, pt.PT_Inversion)); oki.append(1)
    except ValueError:
        ti.append(np.zeros(100)); oki.append(0)
out["inv_params"], out["inv_T"], out["inv_ok"] = pi_, np.array(ti), np.array(oki)
pa = np.column_stack([rng.uniform(1000, 2000, 8), rng.uniform(1.2, 1.67, 8), rng.uniform(-1, 1.5, 8)])
out["adiab_params"] = pa
out["adiab_T"] = np.array([pt.PT_generator(p, q, pt.PT_adiabatic) for q in pa])
pp = np.column_stack([rng.uniform(1200, 2000, 8)] + [rng.uniform(0, 300, 8) for _ in range(7)])
out["piette_params"] = pp
out["piette_T"] = np.array([pt.PT_generator(p, q, pt.PT_piette) for q in pp])
np.savez_compressed(os.path.join(HERE, "pt_golden.npz"), **out)

# ---------------------------------------------------------------- Planck, radpress
wn_s = np.array([500.0, 1000.0, 2500.0, 5000.0, 11000.0])
T_s = np.array([400.0, 1000.0, 1424.5564977665601, 3000.0])
misc["planck"] = {"wn": wn_s.tolist(), "T": T_s.tolist(), "B": cf.Planck(T_s, wn_s).tolist()}
press = np.logspace(-5, 2, 100)   # radpress takes top->bottom... (pressure ascending index = file order before reformat)
temp = 1100 + 500 * (np.log10(press) + 5) / 7
mu = np.full(100, 2.3) + 0.05 * np.sin(np.arange(100))
rad = mat.radpress(press, temp, mu, 0.1, rplanet / 1000.0, sc.G * mplanet / rplanet ** 2)
rad2 = mat.radpress(press, temp, mu, 0.123, rplanet / 1000.0, sc.G * mplanet / rplanet ** 2)
misc["radpress"] = {"press_bar": press.tolist(), "temp": temp.tolist(), "mu": mu.tolist(),
                    "p0_bar": 0.1, "R0_km": rplanet / 1000.0, "g0_ms2": sc.G * mplanet / rplanet ** 2,
                    "rad_km": rad.tolist(), "p0b_bar": 0.123, "radb_km": rad2.tolist()}

# ---------------------------------------------------------------- atm file round trip
tmp = "/tmp/bartrt_golden"
os.makedirs(tmp, exist_ok=True)
case = synth.make_case(tmp, nlayers=100, nwave=16, opmol=(), cia=False)
sp_, pr_, te_, ab_ = mat.readatm(case.atm)
misc["readatm"] = {"species": list(sp_), "press0": float(pr_[0]), "press99": float(pr_[-1]),
                   "temp0": float(te_[0]), "temp99": float(te_[-1]),
                   "abund_row0": ab_[0].tolist(), "press_sum": float(pr_.sum()),
                   "temp_sum": float(te_.sum())}

# ---------------------------------------------------------------- filters / star / band integration
synth.blackbody_kurucz(os.path.join(tmp, "star.pck"))
starfl, starwn, tmodel, gmodel = w.readkurucz(os.path.join(tmp, "star.pck"), tstar, gstar)
misc["kurucz"] = {"tmodel": float(tmodel), "gmodel": float(gmodel),
                  "starfl_sum": float(starfl.sum()), "starwn0": float(starwn[0]),
                  "n": int(len(starwn))}
wout = {}
sets = {
    "demo": (2500.0 + np.arange(2501), [os.path.join(REF, "inputs/filters/demo/fdemo%02d.dat" % i)
                                        for i in range(1, 11)]),
    "irac": (910.0 + np.arange(2424), [os.path.join(REF, "inputs/filters/spitzer_irac%d_fa.dat" % i)
                                       for i in range(1, 5)]),
}
for name, (specwn, files) in sets.items():
    idx0, npts, nif, ist = [], [], [], []
    for k, f in enumerate(files):
        fwn, ftr = w.readfilter(f)
        # the filter curve itself: the file's two numeric columns (input data) and what
        # the reference's readfilter makes of them, so that hostio.readfilter / resample
        # are held to the reference on boxes without /root/reference too
        wout["%s_filter%02d_columns" % (name, k)] = np.loadtxt(f, comments="#")
        wout["%s_filter%02d_wn" % (name, k)] = fwn
        wout["%s_filter%02d_tr" % (name, k)] = ftr
        a, b, ind = w.resample(specwn, fwn, ftr, starwn, starfl)
        ind = ind[0]
        assert np.all(np.diff(ind) == 1)
        idx0.append(ind[0]); npts.append(len(ind)); nif.append(a); ist.append(b)
    wout[name + "_specwn"] = specwn
    wout[name + "_idx0"] = np.array(idx0); wout[name + "_npts"] = np.array(npts)
    wout[name + "_nifilter"] = np.concatenate(nif); wout[name + "_istarfl"] = np.concatenate(ist)
    spectra = 1e4 * (1 + rng.random((3, len(specwn)))) * (specwn / specwn[0]) ** 1.5
    rprs = rplanet / rstar
    band = np.zeros((3, len(files))); band_direct = np.zeros((3, len(files)))
    for s in range(3):
        off = 0
        for i in range(len(files)):
            ind = (np.arange(idx0[i], idx0[i] + npts[i]),)
            fluxrat = (spectra[s][ind] / ist[i]) * rprs * rprs
            band[s, i] = w.bandintegrate(fluxrat, specwn, nif[i], ind)
            band_direct[s, i] = w.bandintegrate(spectra[s][ind], specwn, nif[i], ind)
    wout[name + "_spectra"] = spectra
    wout[name + "_band_eclipse"] = band
    wout[name + "_band_direct"] = band_direct
wout["starwn"], wout["starfl"] = starwn, starfl
wout["rprs"] = np.array(rplanet / rstar)
np.savez_compressed(os.path.join(HERE, "wine_golden.npz"), **wout)

# ---------------------------------------------------------------- abundance renormalisation
species = np.array(["He", "H2", "CO", "CO2", "CH4", "H2O"])
abund = np.tile([0.15, 0.85, 1e-4, 1e-4, 1e-4, 1e-4], (100, 1)) * (1 + 0.01 * rng.random((100, 6)))
iH2 = np.where(species == "H2")[0]; iHe = np.where(species == "He")[0]
ratio = (abund[:, iH2] / abund[:, iHe]).squeeze()
imetals = np.where((species != "He") & (species != "H2") & (species != "H-") & (species != "e-"))[0]
imol = np.array([2, 3, 4, 5])
fac = np.vstack([rng.uniform(-2, 1, (6, 4)), [[4.1, 0, 0, 0]]])   # last one drives q < 0
res, bad = [], []
for f in fac:
    ap = abund.T.copy()
    for i, m in enumerate(imol):
        ap[m] = abund[:, m] * 10.0 ** f[i]
    q = 1.0 - np.sum(ap[imetals], axis=0)
    bad.append(bool(np.any(q < 0.0)))
    ap[iH2] = ratio * q / (1.0 + ratio)
    ap[iHe] = q / (1.0 + ratio)
    res.append(ap)
np.savez_compressed(os.path.join(HERE, "abund_golden.npz"), abund=abund, fac=fac, imol=imol,
                    result=np.array(res), bad=np.array(bad), species=species)

json.dump(misc, open(os.path.join(HERE, "misc_golden.json"), "w"), indent=1)
print("golden vectors written to", HERE)
