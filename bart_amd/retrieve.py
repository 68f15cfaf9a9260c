"""Run a retrieval: the batched sampler driving the GPU worker.

    python -m bart_amd.retrieve -c BART.cfg [--out DIR]
    python -m torch.distributed.run --nproc-per-node 8 -m bart_amd.retrieve -c BART.cfg

The second form shards the wavenumber axis over the node's GPUs (one process
per GPU; every rank runs the same seeded sampler, each computes its block of
every spectrum, one RCCL all-gather per step reassembles them).  Reads the
reference's ``[MCMC]`` keys (examples/demo/BART_eclipse.cfg) and writes
``output.npy`` (posterior sample [nchains, nsteps, npars]), ``bestFit.txt`` and
``MCMC.log`` in the output directory.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np

from . import BARTfunc, sampler


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config_file", required=True)
    ap.add_argument("--out", default=None)
    ap.add_argument("--numit", type=int, default=None)
    ap.add_argument("--python-loop", action="store_true",
                    help="run the sampler loop in Python (sampler.run) instead of the native one")
    a = ap.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    wcfg = BARTfunc.WorkerConfig.from_cfg(a.config_file)
    scfg = sampler.SamplerConfig.from_cfg(a.config_file)
    if a.numit:
        scfg.numit = a.numit
    w = BARTfunc.Worker(wcfg, shard=(rank, world) if world > 1 else None, device=local)
    out = a.out or os.path.dirname(os.path.abspath(a.config_file))
    lines = []

    def log(msg):
        lines.append(msg)
        if rank == 0:
            print(msg, flush=True)

    t0 = time.perf_counter()
    native = not a.python_loop and world == 1
    res = sampler.run_native(w, scfg, log=log) if native else sampler.run(w.step, scfg, log=log)
    dt = time.perf_counter() - t0
    nmodel = res["chain"].shape[0] * res["chain"].shape[1]
    log("%d models in %.2f s (%.0f models/s); acceptance %.3f; best chisq %.4f" % (
        nmodel, dt, nmodel / dt, res["accept_rate"], res["best_chisq"]))
    if res["grstat"] is not None:
        log("Gelman-Rubin: " + " ".join("%.3f" % g for g in res["grstat"]))
    log("Bad iterations due to temperature %d, abundance %d, energy %d" % (
        w.nbad[1], w.nbad[2], w.nbad[3]))
    if rank == 0:
        os.makedirs(out, exist_ok=True)
        np.save(os.path.join(out, "output.npy"), res["chain"])
        with open(os.path.join(out, "bestFit.txt"), "w") as f:
            f.write("# best-fit parameters, chisq = %.6f\n" % res["best_chisq"])
            f.write(" ".join("%.8g" % p for p in res["bestp"]) + "\n")
        with open(os.path.join(out, "MCMC.log"), "w") as f:
            f.write("\n".join(lines) + "\n")
    w.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
