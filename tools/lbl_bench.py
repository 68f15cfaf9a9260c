"""BASELINE.json config 5: on-the-fly Voigt line-by-line spectrum (no opacity
table), ~1e6 synthetic lines on a 1e5-point grid, 100 layers, one walker.
Prints one JSON line with seconds per spectrum and line-layer pairs per second."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import engine, synth_lbl, transit_module as trm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lines", type=int, default=250000, help="lines per molecule (4 molecules)")
ap.add_argument("--nwave", type=int, default=100000)
ap.add_argument("--nlayers", type=int, default=100)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ptop", type=float, default=1e-5, help="top pressure, bar")
ap.add_argument("--pbottom", type=float, default=100.0, help="bottom pressure, bar")
a = ap.parse_args()

d = os.path.join(tempfile.gettempdir(), "bartrt_lbl_bench_%g_%g" % (a.ptop, a.pbottom))
mols = ("H2O", "CO", "CO2", "CH4")
t0 = time.perf_counter()
case = synth_lbl.make_lbl_case(d, molecules=mols, nlines=a.lines, nwave=a.nwave, wnlow=1000.0,
                               wndelt=0.1, nlayers=a.nlayers, cia=True, ptop=a.ptop,
                               pbottom=a.pbottom)
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
assert np.all(np.isfinite(spec)) and spec.min() >= 0
best = min(ts)
print(json.dumps({
    "workload": "on-the-fly Voigt line-by-line, %d lines (4 molecules x 2 isotopologues), "
                "%d-point grid, %d layers, nwidth 20, ethresh 1e-6" % (4 * a.lines, n, a.nlayers),
    "seconds_per_spectrum": best, "all_runs_s": ts, "init_s": t_init, "input_generation_s": t_gen,
    "line_layer_pairs_per_s": 4 * a.lines * a.nlayers / best,
    "spectrum_min": float(spec.min()), "spectrum_max": float(spec.max())}))
trm.free_memory()
