"""BASELINE.json config 2: demo eclipse shape (CH4 table, 2-4 um = 2501 samples,
100 layers), ONE walker through the reference-shaped call
trm.run_transit(profiles.flatten(), nwave) with host buffers (PCIe included)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import engine, synth, transit_module as trm  # noqa: E402

d = os.path.join(tempfile.gettempdir(), "bartrt_demo_latency")
case = synth.make_case(d, nlayers=100, nwave=2501, wnlow=2500.0, opmol=("CH4",), seed=7)
trm.transit_init(3, ["transit", "-c", case.tcfg])
n = trm.get_no_samples()
prof = case.profiles().ravel()
for _ in range(20):
    trm.run_transit(prof, n)
engine.walked_begin(); trm.run_transit(prof, n); kname = engine.walked_end()[2]
engine.timing_begin(1)
ts = []
for _ in range(300):
    t0 = time.perf_counter()
    s = trm.run_transit(prof, n)
    ts.append(time.perf_counter() - t0)
ms, nl = engine.timing_end()
ts = np.array(ts) * 1e6
print(json.dumps({"workload": "demo eclipse shape, 1 walker, host buffers, run_transit()",
                  "kernel": kname, "rt_kernel_us": ms / max(nl, 1) * 1e3, "nwave": n, "median_us": float(np.median(ts)), "p10_us": float(np.percentile(ts, 10)),
                  "p90_us": float(np.percentile(ts, 90)), "spectra_per_s": float(1e6 / np.median(ts))}))
trm.free_memory()
