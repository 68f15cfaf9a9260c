#!/usr/bin/env python3
"""End-to-end example on synthetic inputs shaped like the reference's demo
(examples/demo/BART_eclipse.cfg: CH4, 2-4 um, 10 band-passes, PT_line):

  1. write a seeded input set (atmosphere, CH4 opacity grid, CIA, TEP, stellar
     model, filters, MCMC configuration) into a work directory,
  2. compute the "observed" eclipse depths from known parameters,
  3. retrieve them back with the batched sampler (all chains per GPU call),
  4. print truth, posterior mean and width per fitted parameter.

    python examples/demo_retrieval.py [workdir] [--numit 200000] [--nchains 10]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import BARTfunc, sampler, synthcfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workdir", nargs="?", default=os.path.join(tempfile.gettempdir(), "bart_amd_demo"))
    ap.add_argument("--numit", type=int, default=200000)
    ap.add_argument("--nchains", type=int, default=10)
    a = ap.parse_args()
    names = ["log kappa", "log g1", "log g2", "alpha", "beta", "log CH4"]
    truth = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5])          # BART_eclipse.cfg:80
    case, cfg = synthcfg.make_worker_case(a.workdir, params=tuple(truth))
    worker = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        depths = worker.step(truth)[0]
        print("eclipse depths in the %d bands: %s" % (worker.nfilters, " ".join("%.3e" % d for d in depths)))
        scfg = sampler.SamplerConfig(
            params=truth.copy(),
            pmin=np.array([-5.0, -2.0, -2.0, 0.0, 0.55, -9.0]),      # BART_eclipse.cfg:81-82
            pmax=np.array([-1.0, 1.0, 1.0, 1.0, 1.2, 1.5]),
            stepsize=np.array([0.01, 0.01, 0.0, 0.0, 0.001, 0.1]),   # :83
            data=depths, uncert=0.02 * depths, nchains=a.nchains, numit=a.numit,
            burnin=a.numit // a.nchains // 5, walk="snooker", seed=1)
        t0 = time.perf_counter()
        res = sampler.run_native(worker, scfg, log=print)
        dt = time.perf_counter() - t0
        post = res["chain"][:, scfg.burnin:, :].reshape(-1, len(truth))
        print("%d forward models in %.2f s (%.0f per second)" % (a.numit, dt, a.numit / dt))
        print("%-10s %10s %12s %10s" % ("parameter", "truth", "mean", "sd"))
        for j in res["free"]:
            print("%-10s %10.4f %12.4f %10.4f" % (names[j], truth[j], post[:, j].mean(), post[:, j].std()))
        print("rejected models: temperature %d, abundance %d" % (worker.nbad[1], worker.nbad[2]))
    finally:
        worker.close()


if __name__ == "__main__":
    main()
