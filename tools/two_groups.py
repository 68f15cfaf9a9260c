"""Independent chain groups on one GPU: G processes, each stepping its own batch of ten walkers through
engine.run_batch_dev back to back (the headline workload of bench.py), started together.  A ten-walker launch leaves
478 of the 1 024 SIMDs idle from half-time on (MEASUREMENTS_ARCHIVE.md, "columns that migrate"): launches of ANOTHER process fill
them, which one process's own launches -- ordered on its stream -- cannot.  A chain group needs its step's results before
it proposes the next, so this is not the headline's metric (one group, `value` of bench.py); it is what a user running
several independent groups (BART's `nchains` split over runs, or several retrievals) gets per GPU.
usage (GPU box): python tools/two_groups.py [groups ...]      e.g. 1 2 3 4   -> one JSON line
(child: python tools/two_groups.py --child <tcfg> <walkers> <steps>)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tcfg, n, steps):
    import numpy as np
    import torch
    import bench
    from bart_amd import engine, synth, transit_module as trm
    case = synth.make_case(os.path.dirname(tcfg), kappa_model="survey8d", reuse=True)
    engine.init(case.tcfg)
    nw = trm.get_no_samples()
    profs = bench.make_profiles(case, n * 8, seed=3 + os.getpid() % 7).reshape(8, n, -1)
    d = torch.from_numpy(profs).cuda()
    out = torch.empty((n, nw), dtype=torch.float64, device="cuda")
    for i in range(40):
        engine.run_batch_dev(d[i % 8], out)
    torch.cuda.synchronize()
    print("ready", flush=True)
    assert sys.stdin.readline().strip() == "go"
    t0 = time.perf_counter()
    for i in range(steps):
        engine.run_batch_dev(d[i % 8], out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("done " + json.dumps({"steps": steps, "walkers": n, "seconds": dt, "us_per_step": dt / steps * 1e6}), flush=True)
    sys.stdin.readline()
    trm.free_memory()


def main():
    from bart_amd import synth
    wd = os.path.join(tempfile.gettempdir(), "bartrt_bench_single_survey8d")
    case = synth.make_case(wd, kappa_model="survey8d", reuse=True)
    rep = {"workload": "ten walkers per step and group, headline shape (100 layers x 1e4 samples, 4 molecules), each group its own process and engine, steps queued back to back"}
    for g in [int(x) for x in (sys.argv[1:] or ["1", "2", "3"])]:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", case.tcfg, "10", "600"], stdin=subprocess.PIPE,
                               stdout=subprocess.PIPE, text=True, env=dict(os.environ, BARTRT_SHARE_OPACITY="0")) for _ in range(g)]
        for p in ps:
            assert p.stdout.readline().strip() == "ready"
        for p in ps:
            p.stdin.write("go\n"); p.stdin.flush()
        res = [json.loads(p.stdout.readline().split(" ", 1)[1]) for p in ps]
        for p in ps:
            p.stdin.write("bye\n"); p.stdin.flush(); p.wait(timeout=60)
        slow = max(r["seconds"] for r in res)
        rep["groups_%d" % g] = {"aggregate_spectra_per_s": sum(r["steps"] * r["walkers"] for r in res) / slow,
                                "us_per_step_of_a_group": [round(r["us_per_step"], 1) for r in res]}
    print(json.dumps(rep))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        main()
