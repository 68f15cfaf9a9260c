"""One MC3-style worker process of the N-process drop-in path: the reference runs
one BARTfunc worker per chain (examples/demo/BART_eclipse.cfg:90-91), each with its own transit instance --
transit_init, then run_transit once per MCMC step.  This child does exactly that through bart_amd.transit_module
and reports what it measured as one JSON line.  Started by tools/bench_configs.py (mc3_processes) and by
tests/test_gpu_share.py; talks to its parent over stdin / stdout:
    child  -> "ready {json}"     after transit_init
    parent -> "go"               once every child is ready
    child  -> "done {json}"      after its steps
    parent -> "bye"              once every child is done (the grid's owner must outlive the others' use)
usage: mc3_child.py <transit.cfg> <rank> <nsteps> [<out.npy>]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    tcfg, rank, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    from bart_amd import transit_module as trm
    t0 = time.perf_counter()
    trm.transit_init(3, ["transit", "-c", tcfg])
    t_init = time.perf_counter() - t0
    n = trm.get_no_samples()
    shared, owner = trm.get_share()
    prof0 = np.zeros(trm.lib().bartrt_get_nprof())
    trm.check(trm.lib().bartrt_get_atm_profile(trm._ptr(prof0), prof0.size))
    L = trm.lib().bartrt_get_nlayers()
    # this chain's walker: the atmosphere file's profile with the temperatures shifted by the rank
    mine = prof0.copy()
    mine[:L] = np.clip(mine[:L] + 20.0 * rank, 410.0, 2990.0)
    trm.run_transit(mine, n)                       # first call (workspaces)
    print("ready " + json.dumps({"rank": rank, "init_s": t_init, "shared": shared, "owner": owner}), flush=True)
    assert sys.stdin.readline().strip() == "go"
    t0 = time.perf_counter()
    for _ in range(nsteps):
        spec = trm.run_transit(mine, n)
    dt = time.perf_counter() - t0
    common = trm.run_transit(prof0, n)
    if out:
        np.save(out, np.stack([common, spec]))
    print("done " + json.dumps({"rank": rank, "steps": nsteps, "loop_s": dt, "us_per_step": dt / max(nsteps, 1) * 1e6}), flush=True)
    sys.stdin.readline()
    trm.free_memory()


if __name__ == "__main__":
    main()
