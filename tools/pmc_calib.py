"""FETCH_SIZE calibration on a known byte count in the RT kernel's own access
pattern (the table layout's 16-byte loads per lane, every byte read once): ONE walker, toomuch = 1e30
(no early exit) brings 2*L*M*W*8 bytes of the opacity grid in from HBM (every layer's two planes,
each once) + the few distinct CIA pair planes its temperatures bracket (the other layers re-read
them from L2) + its records, and writes one spectrum.
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
trm_cia = trm.get_cia_interp()
prof = case.profiles().ravel()[None, :]
for _ in range(5):
    engine.run_batch(prof)
L, M, W = 100, 4, 10000
t = case.temp0
ct = np.arange(400.0, 3000.1, 200.0)                      # the synthetic H2-H2 file's temperatures (bart_amd/synth.py)
pairs = np.unique(np.clip(np.searchsorted(ct, np.clip(t, ct[0], ct[-1]), side="right") - 1, 0, len(ct) - 2))
slots = 2 if trm_cia == "spline" else 1                   # values (+ second derivatives in T): one pair plane each
expected = 2 * L * M * W * 8 + slots * len(pairs) * W * 16 + L * (4 + 2 * M + 2 * slots + 1 + slots) * 8 + W * 8
print("expected HBM bytes per launch:", expected)
trm.free_memory()
