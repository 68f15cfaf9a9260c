"""BASELINE.json config 5: on-the-fly Voigt line-by-line spectrum (no opacity
table), ~1e6 synthetic lines on a 1e5-point grid, 100 layers, one walker.

    python tools/lbl_bench.py [--wnosamp 1|2160] [--lines N] [--nwave W] ...

Prints one JSON line: seconds per spectrum, line-layer pairs per second and the
work unit of SURVEY.md 8d for this path -- per (kept line, layer) the Voigt profile
samples inside its cut, 2 * nwidth * HWHM / (wndelt / dv), summed over the run
(`voigt_samples`), with the bytes the line list and the extinction array account
for.  The path is arithmetic: one profile sample costs ~60-150 fp64 operations
(rational Faddeeva approximation), the bytes per sample are a fraction of one.
`bench.py --config lbl` prints the same line."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

H, LS, KB, AMU = 6.6260755e-27, 2.99792458e10, 1.380658e-16, 1.66053886e-24
MOLS = ("H2O", "CO", "CO2", "CH4")


# fp64 VALU operations of one Voigt sample by branch of voigt_k (csrc/lbl.hip; counted in the gfx950 assembly of
# lbl_accumulate_fine, the function value and its accumulation, without the step's bookkeeping): far wing
# |z| >= 100 (three series terms as a polynomial in one variable), 17 <= |z| < 100 (six terms), 8 <= |z| < 17
# (eleven), |z| < 8: Weideman N = 36 for y > 0.13, else the expansion about the real axis (order 2-10 by y)
VOIGT_OPS = {"far": 19, "series6": 32, "series11": 42, "weideman": 98, "taylor0": 42, "taylor_per_order": 7}
TAYLOR_YMAX = 0.13
TAYLOR_STEPS = np.array([6.0e-6, 5.0e-4, 3.0e-3, 9.5e-3, 2.2e-2, 4.0e-2, 6.5e-2, 9.5e-2])


def work_units(case, nwidth, ethresh, wnosamp):
    """Kept (line, layer) pairs and Voigt samples inside their cuts (numpy; the widths
    of scripts/broadening.py:121-143 and the sampling rule of DESIGN.md C15)."""
    from bart_amd import synth
    press = case.press_bar * 1e6
    T = case.temp0
    q = case.abund0
    sp = case.species
    mass = np.array([synth.MOLECULES[s][1] for s in sp])
    diam = np.array([synth.MOLECULES[s][2] for s in sp]) * 1e-8
    ih2, ihe = sp.index("H2"), sp.index("He")
    wndelt = case.wn[1] - case.wn[0]
    divs = [d for d in range(1, wnosamp + 1) if wnosamp % d == 0]
    kept = samples = 0.0
    branch = {"far": 0.0, "mid": 0.0, "core": 0.0}   # by branch of voigt_k (csrc/lbl.hip): |z| >= 100, 8 .. 100, < 8
    ops = 0.0                                        # fp64 VALU operations of those samples (VOIGT_OPS below)
    dvs = []
    for l in range(len(press)):
        per_db, wmin = [], np.inf
        for db in case.linedbs:
            s = sp.index(db["molecule"])
            nu0, iso = np.asarray(db["wn"]), np.asarray(db["iso"])
            S, hw = np.zeros(len(nu0)), np.zeros(len(nu0))
            aD, aLv = np.zeros(len(nu0)), np.zeros(len(nu0))
            for i, info in enumerate(db["isotopes"]):
                m = iso == i
                mi = info["mass"] * AMU
                Z = np.interp(T[l], db["temps"], info["Z"])
                S[m] = (info["ratio"] * q[l, s] / Z * np.asarray(db["gf"])[m]
                        * np.exp(-H * LS / KB * np.asarray(db["elow"])[m] / T[l])
                        * (1 - np.exp(-H * LS / KB * nu0[m] / T[l])))
                dop = np.sqrt(2 * np.log(2) * KB * T[l] / mi) / LS
                col = sum(q[l, c] * (0.5 * (diam[s] + diam[c])) ** 2 * np.sqrt(1 / mi + 1 / (mass[c] * AMU))
                          for c in (ih2, ihe))
                aL = np.sqrt(2.0) / (LS * np.sqrt(np.pi * KB * T[l])) * press[l] * col
                hw[m] = np.maximum(nu0[m] * dop, aL)
                aD[m], aLv[m] = nu0[m] * dop, aL
                wmin = min(wmin, max(case.wn[0] * dop, aL))
            per_db.append((S, hw, aD, aLv))
        dv = next((d for d in divs if wndelt / d <= 0.5 * wmin), wnosamp) if wnosamp > 1 else 1
        dvs.append(dv)
        for S, hw, aD, aLv in per_db:
            keep = (S >= ethresh * S.max()) & (S > 0)
            kept += keep.sum()
            step = wndelt / dv
            cut = nwidth * hw[keep]
            samples += (2 * cut / step).sum()
            # z = sqrt(ln 2) ((nu - nu0) + i alpha_L) / alpha_D: half-widths in wavenumber of |z| < 8 and < 100
            sc = aD[keep] / np.sqrt(np.log(2.0))
            y2 = (aLv[keep] / sc) ** 2
            in8 = np.minimum(cut, sc * np.sqrt(np.maximum(64.0 - y2, 0.0)))
            in100 = np.minimum(cut, sc * np.sqrt(np.maximum(1e4 - y2, 0.0)))
            in17 = np.minimum(cut, sc * np.sqrt(np.maximum(289.0 - y2, 0.0)))
            branch["core"] += (2 * in8 / step).sum()
            branch["mid"] += (2 * (in100 - in8) / step).sum()
            branch["far"] += (2 * (cut - in100) / step).sum()
            yy = np.sqrt(y2)
            order = np.where(yy <= TAYLOR_YMAX, 2 + (yy[:, None] > TAYLOR_STEPS[None, :]).sum(axis=1), 0)
            core_ops = np.where(order > 0, VOIGT_OPS["taylor0"] + VOIGT_OPS["taylor_per_order"] * (order - 1),
                                VOIGT_OPS["weideman"])
            ops += (2 * in8 / step * core_ops).sum() + VOIGT_OPS["series11"] * (2 * (in17 - in8) / step).sum() \
                + VOIGT_OPS["series6"] * (2 * (in100 - in17) / step).sum() + VOIGT_OPS["far"] * (2 * (cut - in100) / step).sum()
    return kept, samples, dvs, branch, ops


def run(argv=None):
    from bart_amd import engine, synth_lbl, transit_module as trm
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=250000, help="lines per molecule (4 molecules)")
    ap.add_argument("--nwave", type=int, default=100000)
    ap.add_argument("--nlayers", type=int, default=100)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--wnosamp", type=int, default=1,
                    help="oversampling of the line sums (reference cfgs: 2160); 1 = on the output points")
    ap.add_argument("--ptop", type=float, default=1e-5, help="top pressure, bar")
    ap.add_argument("--pbottom", type=float, default=100.0, help="bottom pressure, bar")
    a = ap.parse_args(argv)

    d = os.path.join(tempfile.gettempdir(), "bartrt_lbl_bench_%g_%g_%d" % (a.ptop, a.pbottom, a.wnosamp))
    t0 = time.perf_counter()
    case = synth_lbl.make_lbl_case(d, molecules=MOLS, nlines=a.lines, nwave=a.nwave, wnlow=1000.0,
                                   wndelt=0.1, nlayers=a.nlayers, cia=True, ptop=a.ptop,
                                   pbottom=a.pbottom, wnosamp=a.wnosamp)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    engine.init(case.tcfg)
    t_init = time.perf_counter() - t0
    n = trm.get_no_samples()
    prof = case.profiles().ravel()
    spec = trm.run_transit(prof, n)            # warm-up
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        spec = trm.run_transit(prof, n)
        ts.append(time.perf_counter() - t0)
    # (finite, not positive: under the default integration rule (1, App. A-4) a line core that puts a
    # tau step >> 1 on one layer can drive a sample negative; rule 0 keeps every sample positive)
    assert np.all(np.isfinite(spec))
    best = min(ts)
    integ = trm.get_integ()
    trm.free_memory()
    kept, samples, dvs, branch, ops = work_units(case, 20.0, 1e-6, a.wnosamp)
    pairs = 4 * a.lines * a.nlayers
    res = {
        "metric": "line-by-line spectra/sec (1e6 lines x 1e5 wavenumbers x 100 layers, BASELINE config 5)",
        "value": 1.0 / best, "unit": "spectra/s",
        "workload": "on-the-fly Voigt line-by-line, %d lines (4 molecules x 2 isotopologues), "
                    "%d-point grid, %d layers, nwidth 20, ethresh 1e-6, wnosamp %d (per-layer factors %d..%d)"
                    % (4 * a.lines, n, a.nlayers, a.wnosamp, min(dvs), max(dvs)),
        "seconds_per_spectrum": best, "all_runs_s": ts, "init_s": t_init, "input_generation_s": t_gen,
        "line_layer_pairs_per_s": pairs / best,
        "kept_line_layer_pairs": kept, "voigt_samples": samples, "voigt_samples_per_s": samples / best,
        "voigt_samples_by_branch": branch, "voigt_fp64_ops": ops, "voigt_ops_model": VOIGT_OPS, "integ": integ,
        # 24 B per line per layer (wavenumber, E_low, gf; + 4 B isotope id here) + the
        # extinction array written once and read once by the RT kernel
        "algorithmic_bytes": 28.0 * pairs + 2 * 8.0 * a.nlayers * n,
        "algorithmic_GBps": (28.0 * pairs + 2 * 8.0 * a.nlayers * n) / best / 1e9,
        "bytes_per_voigt_sample": (28.0 * pairs + 2 * 8.0 * a.nlayers * n) / max(samples, 1),
        "spectrum_min": float(spec.min()), "spectrum_max": float(spec.max())}
    print(json.dumps(res))
    return res


if __name__ == "__main__":
    run()
