"""PCIe-inclusive rate of the C-ABI's host-buffer entry point (bartrt_run_transit_batch:
profiles in from host memory, spectra out to host memory, one synchronous call per batch) on
the bench grid -- DESIGN.md section 6; never the bench's `value`.
    python tools/pcie_rate.py [walkers ...]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bart_amd import engine, synth, transit_module as trm  # noqa: E402

case = synth.make_case(os.path.join(tempfile.gettempdir(), "bartrt_bench_100x10000"), nlayers=100, nwave=10000, reuse=True)
engine.init(case.tcfg)
for nw in [int(x) for x in sys.argv[1:]] or [1, 10, 64, 256]:
    profs = bench.make_profiles(case, nw * 4, seed=5).reshape(4, nw, -1)
    for i in range(5):
        engine.run_batch(profs[i % 4])
    n = max(20, 2000 // nw)
    t0 = time.perf_counter()
    for i in range(n):
        spec = engine.run_batch(profs[i % 4])
    dt = (time.perf_counter() - t0) / n
    assert spec.shape == (nw, 10000) and np.all(np.isfinite(spec))
    print(json.dumps({"workload": "bartrt_run_transit_batch, host buffers in and out, 100 layers x 1e4 wavenumbers",
                      "walkers": nw, "us_per_call": dt * 1e6, "spectra_per_s": nw / dt,
                      "bytes_in_per_call": int(profs[0].nbytes), "bytes_out_per_call": int(spec.nbytes)}))
trm.free_memory()
