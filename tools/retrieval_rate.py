"""Iterations per second of the in-process batched sampler (DEMC / snooker, all
chains evaluated by one Worker.step call per iteration) on the WASP-12b
retrieval shape (BASELINE config 4's problem on one GPU)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import BARTfunc, sampler, synthcfg  # noqa: E402

mols = ("H2O", "CO", "CO2", "CH4")
truth = np.array([-1.5, -0.8, -0.8, 0.5, 1.0, -0.3, 0.2, -0.5, 0.1])
d = os.path.join(tempfile.gettempdir(), "bartrt_retrate")
case, cfg = synthcfg.make_worker_case(d, nwave=2424, wnlow=910.0, opmol=mols, molfit=mols,
                                      params=tuple(truth), nfilters=4, reuse=True)
w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
data = w.step(truth)[0]
for nch in (10, 32):
    numit = 400 * nch
    scfg = sampler.SamplerConfig(
        params=truth + 0.02, pmin=np.array([-5, -2, -2, 0, 0.55, -9, -9, -9, -9.0]),
        pmax=np.array([-1, 1, 1, 1, 1.2, 1.5, 1.5, 1.5, 1.5]),
        stepsize=np.array([0.01, 0.01, 0.01, 0.01, 0.001, 0.05, 0.05, 0.05, 0.05]),
        data=data, uncert=data * 0.01, nchains=nch, numit=numit, burnin=50, walk="snooker", seed=1)
    for name, fn in (("python loop", lambda: sampler.run(w.step, scfg)),
                     ("native loop", lambda: sampler.run_native(w, scfg))):
        fn()                                       # warm-up
        t0 = time.perf_counter()
        res = fn()
        dt = time.perf_counter() - t0
        print(json.dumps({"workload": "WASP-12b shape (100 layers x 2424 samples, 4 molecules, 4 filters), "
                                      "snooker DEMC, %d chains, %s" % (nch, name),
                          "iterations_per_s": round(numit / nch / dt, 1),
                          "model_evaluations_per_s": round(numit / dt),
                          "acceptance": round(res["accept_rate"], 3)}))
w.close()
