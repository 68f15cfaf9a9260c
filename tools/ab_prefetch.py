"""A/B of prefetched preparation (bartrt_prefetch_profiles_dev) on the bench grid: step time
with the next batch's prep_profiles carried by the current RT launch, against the plain sequence.
usage: python tools/ab_prefetch.py [walkers ...]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bart_amd import engine, synth, transit_module as trm  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:]] or [1, 10, 64]
    case = synth.make_case(os.path.join(tempfile.gettempdir(), "bartrt_bench_single"), reuse=True)
    engine.init(case.tcfg)
    for n in batches:
        nsets = 16
        profs = bench.make_profiles(case, n * nsets, seed=20260103).reshape(nsets, n, -1)
        d_prof = torch.from_numpy(profs).cuda()
        outs = [torch.empty((n, 10000), dtype=torch.float64, device="cuda") for _ in range(2)]
        ref = None
        for on in (False, True, False, True):
            steps = max(50, min(400, 4000 // n))
            nxt = (lambda i: d_prof[(i + 1) % nsets]) if on else (lambda i: None)
            for i in range(10):
                engine.run_batch_dev(d_prof[i % nsets], outs[i & 1], next_prof=nxt(i))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(10, 10 + steps):
                engine.run_batch_dev(d_prof[i % nsets], outs[i & 1], next_prof=nxt(i))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            last = outs[(10 + steps - 1) & 1].clone()
            same = True if ref is None else bool(torch.equal(last, ref))
            ref = last
            print(json.dumps({"walkers": n, "prefetch": on, "us_per_step": round(dt * 1e6, 2), "bit_identical": same}),
                  flush=True)
    trm.free_memory()


if __name__ == "__main__":
    main()
