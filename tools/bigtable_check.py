"""One-off check of a grid above 4 GB (the 32-bit lane offsets and the moving window of
the row-per-layer kernels; the device re-layout of a big table): 100 layers x 27
temperatures x 4 molecules x 52 000 samples = 4.49 GB, eclipse and transit geometry,
1 / 3 / 12 walkers (quad-layer, single-wave, MFMA transit kernels) against the oracle.
    python tools/bigtable_check.py [eclipse|transit|both]         (GPU box; ~2 min, 5 GB of /tmp per geometry)
BARTRT_KERNEL in the environment forces a kernel variant (quad: under the default `cut slant` the one-ray-per-lane
kernel with its moving window; team: the three-wave column team).
Uses the oracle: a test driver, not part of the product."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from bart_amd import engine, synth, transit_module as trm  # noqa: E402
from oracle import rt_oracle as orc  # noqa: E402
from test_gpu_parity import walkers  # noqa: E402

d = os.path.join(tempfile.gettempdir(), "bartrt_bigtable")
worst = 0.0
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, extra in (("eclipse", None), ("transit", {"solution": "transit", "starrad": 1.145})):
    if which not in ("both", name):
        continue
    t0 = time.time()
    c = synth.make_case(os.path.join(d, name), nlayers=100, nwave=52000, extra_keys=extra, reuse=True)
    print(name, "inputs in %.0f s, table %.2f GB" % (time.time() - t0, os.path.getsize(c.opacity) / 1e9), flush=True)
    engine.init(c.tcfg)
    o = orc.OracleEngine(c.tcfg)
    for n in (1, 3, 12):
        p = walkers(c, n, seed=40 + n)
        engine.walked_begin()
        got = engine.run_batch(p)
        kname = engine.walked_end()[2] if name == "eclipse" else "transit"
        ref = o.run_batch(p[:2])
        err = float(np.abs(got[:2] / ref - 1).max())
        worst = max(worst, err)
        print(name, n, "walkers: max rel err %.2e  [%s]" % (err, kname), flush=True)
    trm.free_memory()
    del o
assert worst < 1e-10, worst
print("big-table check ok, worst %.2e" % worst)
