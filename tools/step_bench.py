"""The whole per-step callable on the bench grid:
params[nwalkers][npars] -> T(p) + abundances -> prep + RT -> band fluxes, all on
the device, host only enqueues.  Reports microseconds per step; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.

usage: python tools/step_bench.py [--walkers 10] [--steps 200] [--ebalance]"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bart_amd import BARTfunc, engine, synthcfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walkers", type=int, default=10)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--ebalance", action="store_true")
    a = ap.parse_args()
    mols = ("H2O", "CO", "CO2", "CH4")
    p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
    d = os.path.join(tempfile.gettempdir(), "bartrt_stepbench")
    case, cfg = synthcfg.make_worker_case(d, nwave=10000, wnlow=1000.0, opmol=mols, molfit=mols, params=p0,
                                          nfilters=10, ebalance=a.ebalance, reuse=True)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    rng = np.random.default_rng(5)
    nsets = 16
    pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (nsets, a.walkers, 9))
    pars[..., 3] = np.clip(pars[..., 3], 0, 1)
    d_par = torch.from_numpy(pars).cuda()
    band = None
    for i in range(20):
        band, status = engine.step_batch_dev(d_par[i % nsets], w.nfilters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        band, status = engine.step_batch_dev(d_par[i % nsets], w.nfilters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = int((status.cpu().numpy() == 0).sum())
    print(json.dumps({"workload": "per-step callable, 100 layers x 1e4 wavenumbers, 4 molecules, 10 filters, "
                                  "%d walkers per step%s" % (a.walkers, ", energy balance on" if a.ebalance else ""),
                      "us_per_step": dt / a.steps * 1e6, "walker_steps_per_s": a.walkers * a.steps / dt,
                      "accepted_in_last_batch": ok, "band0": float(band[0, 0])}))
    w.close()


if __name__ == "__main__":
    main()
