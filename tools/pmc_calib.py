"""FETCH_SIZE calibration on a known byte count in the RT kernel's own access
pattern (the table layout's 16-byte loads per lane, every byte read once): ONE walker, toomuch = 1e30
(no early exit) reads exactly 2*L*M*W*8 + 2*L*W*8 bytes of table per launch.
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (MI355X_MICROARCH.md, HBM)."""
import os
import sys
import tempfile

os.environ["BARTRT_KERNEL"] = "mono"  # the kernel the bench line times (one walker would pick the quad-layer one)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import engine, synth, transit_module as trm  # noqa: E402

d = os.path.join(tempfile.gettempdir(), "bartrt_calib")
case = synth.make_case(d, nlayers=100, nwave=10000, toomuch=1e30, reuse=True)
engine.init(case.tcfg)
prof = case.profiles().ravel()[None, :]
for _ in range(5):
    engine.run_batch(prof)
print("expected table bytes per launch:", 2 * 100 * 4 * 10000 * 8 + 2 * 100 * 10000 * 8)
trm.free_memory()
