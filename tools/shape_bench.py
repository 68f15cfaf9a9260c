"""RT launch time on other shapes than the headline one: table molecules, CIA
pairs, layers, samples, geometry.
usage: python tools/shape_bench.py [--nmol 6] [--cia 2] [--layers 100] [--nwave 10000]
                                   [--solution eclipse|transit] [walkers ...]
BARTRT_KERNEL=generic in the environment times the fallback kernel on the same shape."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bart_amd import engine, synth, transit_module as trm  # noqa: E402

MOLS = ("H2O", "CO", "CO2", "CH4", "NH3", "HCN", "C2H2", "TiO", "VO")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nmol", type=int, default=6)
    ap.add_argument("--cia", type=int, default=2)
    ap.add_argument("--layers", type=int, default=100)
    ap.add_argument("--nwave", type=int, default=10000)
    ap.add_argument("--solution", default="eclipse")
    ap.add_argument("--angles", type=int, default=5, help="ray-grid size (evenly spaced over 0 .. 80 degrees)")
    ap.add_argument("--integ", type=int, default=None)
    ap.add_argument("walkers", nargs="*", type=int)
    a = ap.parse_args()
    mols = MOLS[:a.nmol]
    d = os.path.join(tempfile.gettempdir(), "bartrt_shape_%d_%d_%d_%d_%s_%d"
                     % (a.nmol, a.cia, a.layers, a.nwave, a.solution, a.angles))
    extra = {"solution": "transit", "starrad": 1.145} if a.solution == "transit" else None
    grid = (0, 20, 40, 60, 80) if a.angles == 5 else tuple(round(80.0 * i / max(a.angles - 1, 1), 3) for i in range(a.angles))
    case = synth.make_case(d, nlayers=a.layers, nwave=a.nwave, reuse=True, cia=a.cia, opmol=mols,
                           species=("He", "H2") + mols, abund=(0.15, 0.85) + (1e-4,) * a.nmol,
                           extra_keys=extra, raygrid=grid)
    engine.init(case.tcfg)
    if a.integ is not None:
        trm.set_integ(a.integ)
    for n in a.walkers or [1, 10, 256]:
        nsets = 8
        profs = bench.make_profiles(case, n * nsets, seed=11).reshape(nsets, n, -1)
        d_prof = torch.from_numpy(profs).cuda()
        out = torch.empty((n, a.nwave), dtype=torch.float64, device="cuda")
        steps = max(5, min(100, 1000 // n))
        for i in range(3):
            engine.run_batch_dev(d_prof[i % nsets], out)
        torch.cuda.synchronize()
        engine.timing_begin()
        t0 = time.perf_counter()
        for i in range(steps):
            engine.run_batch_dev(d_prof[i % nsets], out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kms, nl = engine.timing_end()
        engine.walked_begin()
        engine.run_batch_dev(d_prof[0], out)
        torch.cuda.synchronize()
        kname = engine.walked_end()[2]
        print(json.dumps({"shape": "%s, %d molecules, %d CIA pairs, %d layers x %d samples, %d ray angles, integ %d"
                          % (a.solution, a.nmol, a.cia, a.layers, a.nwave, a.angles, trm.get_integ()),
                          "kernel": os.environ.get("BARTRT_KERNEL", "default"), "launched": kname, "rtc": trm.get_rtc_stats(), "walkers": n,
                          "spectra_per_s": round(n * steps / dt), "ms_per_step": round(dt / steps * 1e3, 4),
                          "rt_kernel_ms": round(kms / max(nl, 1), 4)}), flush=True)
    trm.free_memory()


if __name__ == "__main__":
    main()
