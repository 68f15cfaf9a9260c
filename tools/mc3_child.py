"""One MC3-style worker process of the N-process drop-in path: the reference runs
one BARTfunc worker per chain (examples/demo/BART_eclipse.cfg:90-91), each with its own transit instance --
transit_init, then run_transit once per MCMC step.  This child does exactly that through bart_amd.transit_module
and reports what it measured as one JSON line.  Started by tools/bench_configs.py (mc3_processes) and by
tests/test_gpu_share.py; talks to its parent over stdin / stdout:
    child  -> "ready {json}"     after transit_init
    parent -> "go"               once every child is ready
    child  -> "done {json}"      after its steps
    parent -> "bye"              once every child is done (the grid's owner must outlive the others' use)
usage: mc3_child.py <transit.cfg> <rank> <nsteps> [<out.npy>] [--radius km] [--cloudtop log10bar]
                    [--scattering flag value] [--until-error] [--late-every K --late-us U [--late-slot S]]
The setters are this process's own trm.set_radius / set_cloudtop / set_scattering (code/BARTfunc.py:350-360);
--until-error: keep calling until the engine refuses (a chain-service client whose owner has died), report it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def kfd_touched() -> bool:
    """Has this process opened the GPU driver (a HIP context exists)?  A chain-service client must not have."""
    try:
        return any("kfd" in os.readlink("/proc/self/fd/" + f) for f in os.listdir("/proc/self/fd"))
    except OSError:
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tcfg"); ap.add_argument("rank", type=int); ap.add_argument("nsteps", type=int)
    ap.add_argument("out", nargs="?")
    ap.add_argument("--radius", type=float); ap.add_argument("--cloudtop", type=float)
    ap.add_argument("--scattering", nargs=2, type=float)
    ap.add_argument("--until-error", action="store_true")
    ap.add_argument("--late-every", type=int, default=0, help="this worker is late for every K-th step ...")
    ap.add_argument("--late-us", type=float, default=0.0, help="... by this many microseconds (a straggler on purpose)")
    ap.add_argument("--late-slot", type=int, default=-1, help="... only if this process holds that slot of the chain service "
                                                              "(slots go by arrival: the straggler is named by slot, not by rank)")
    a = ap.parse_args()
    tcfg, rank, nsteps, out = a.tcfg, a.rank, a.nsteps, a.out
    from bart_amd import transit_module as trm
    t0 = time.perf_counter()
    trm.transit_init(3, ["transit", "-c", tcfg])
    t_init = time.perf_counter() - t0
    n = trm.get_no_samples()
    shared, owner = trm.get_share()
    svc = trm.get_service()
    if a.radius is not None:
        trm.set_radius(a.radius)
    if a.cloudtop is not None:
        trm.set_cloudtop(a.cloudtop)
    if a.scattering is not None:
        trm.set_scattering(int(a.scattering[0]), a.scattering[1])
    prof0 = np.zeros(trm.lib().bartrt_get_nprof())
    trm.check(trm.lib().bartrt_get_atm_profile(trm._ptr(prof0), prof0.size))
    L = trm.lib().bartrt_get_nlayers()
    # this chain's walker: the atmosphere file's profile with the temperatures shifted by the rank
    mine = prof0.copy()
    mine[:L] = np.clip(mine[:L] + 20.0 * rank, 410.0, 2990.0)
    trm.run_transit(mine, n)                       # first call (workspaces)
    print("ready " + json.dumps({"rank": rank, "init_s": t_init, "shared": shared, "owner": owner, "service": svc["mode"],
                                 "slot": svc["slot"], "pid": os.getpid(), "hip_context": kfd_touched()}), flush=True)
    assert sys.stdin.readline().strip() == "go"
    if a.until_error:
        k, err = 0, None
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 60.0:
            try:
                trm.run_transit(mine, n)
                k += 1
            except trm.TransitError as e:
                err = str(e)
                break
        print("done " + json.dumps({"rank": rank, "steps": k, "error": err, "waited_s": time.perf_counter() - t0}), flush=True)
        sys.stdin.readline()
        try:
            trm.free_memory()
        except trm.TransitError:
            pass
        return
    import gc
    gc.collect()          # (before the warm-up calls: the device idles while it runs)
    for _ in range(min(30, nsteps)):            # untimed: the processes fall into step, caches and kernels are warm
        trm.run_transit(mine, n)
    lat = np.zeros(max(nsteps, 1))
    gc.disable()          # (as timeit does: a full collection is a pause of tens of milliseconds in the middle of the loop)
    t0 = time.perf_counter()
    first, differ = None, 0      # the chain's profile is the same every step: so must its spectrum be, bit for bit
    for i in range(nsteps):
        if a.late_every and i % a.late_every == a.late_every - 1 and (a.late_slot < 0 or a.late_slot == svc["slot"]):
            t_late = time.perf_counter() + a.late_us * 1e-6
            while time.perf_counter() < t_late:
                pass
        t1 = time.perf_counter()
        spec = trm.run_transit(mine, n)
        lat[i] = time.perf_counter() - t1
        if a.late_every:             # (not in the timed runs of tools/mc3_bench.py: the compare is 10 us of host time per step)
            if first is None:
                first = spec.copy()
            elif not np.array_equal(first, spec):
                differ += 1
    dt = time.perf_counter() - t0
    gc.enable()
    common = trm.run_transit(prof0, n)
    if out:
        np.save(out, np.stack([common, spec]))
    rep = {"rank": rank, "steps": nsteps, "loop_s": dt, "us_per_step": dt / max(nsteps, 1) * 1e6,
           "call_us_median": float(np.median(lat) * 1e6), "call_us_p90": float(np.percentile(lat, 90) * 1e6),
           "call_us_max": float(lat.max() * 1e6), "calls_over_twice_the_median": int((lat > 2 * np.median(lat)).sum()),
           "slowest_calls": [[int(i), round(float(lat[i]) * 1e6, 1)] for i in np.argsort(lat)[-6:][::-1]],
           "hip_context": kfd_touched(), "steps_that_differ_from_the_first": differ if a.late_every else None}
    if svc["mode"] != "engine":
        rep["service_stats"] = trm.get_service_stats()
    print("done " + json.dumps(rep), flush=True)
    sys.stdin.readline()
    trm.free_memory()


if __name__ == "__main__":
    main()
