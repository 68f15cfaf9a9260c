"""Host-visible time of one Worker.step call (numpy parameters in, band fluxes out:
what an MCMC driver pays per iteration), 1 and 10 walkers, on the bench grid and
on the WASP-12b retrieval shape."""
import os, sys, tempfile, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from bart_amd import BARTfunc, engine, synthcfg
mols = ("H2O", "CO", "CO2", "CH4")
p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
for shape, kw in (("bench grid 1e4", dict(nwave=10000, wnlow=1000.0)), ("WASP shape 2424", dict(nwave=2424, wnlow=910.0))):
    d = os.path.join(tempfile.gettempdir(), "bartrt_hs_%d" % kw["nwave"])
    case, cfg = synthcfg.make_worker_case(d, opmol=mols, molfit=mols, params=p0, nfilters=10 if kw["nwave"] > 5000 else 4, reuse=True, **kw)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    rng = np.random.default_rng(5)
    for n in (1, 10):
        pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (n, 9))
        pars[:, 3] = np.clip(pars[:, 3], 0, 1)
        for _ in range(20): w.step(pars)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); b = w.step(pars); ts.append(time.perf_counter() - t0)
        print(json.dumps({"workload": "Worker.step (host arrays in/out), " + shape, "walkers": n, "median_us": float(np.median(ts) * 1e6)}))
    w.close()
