"""One-off sweep behind tests/test_lbl.py::test_config5_slice_against_the_oracle: blocks of
consecutive samples at several places of the config-5 grid (both ends, where the Doppler
widths differ by a factor eleven), several layers, wnosamp 1 and 2160, against the
scipy-Faddeeva oracle restricted to the block.   python tools/lbl_slice_sweep.py   (GPU box)"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import engine, synth_lbl, transit_module as trm  # noqa: E402
from oracle import lbl_oracle  # noqa: E402

mols = ("H2O", "CO", "CO2", "CH4")
worst = 0.0
for wnosamp, nslice, layers in ((1, 1000, (0, 13, 37, 58, 74, 86, 99)), (2160, 250, (5, 33, 62, 80, 97))):
    d = os.path.join(tempfile.gettempdir(), "lbl_slice_%d" % wnosamp)
    c = synth_lbl.make_lbl_case(d, molecules=mols, nlines=250000, nwave=100000, wnlow=1000.0, wndelt=0.1,
                                nlayers=100, cia=False, wnosamp=wnosamp)
    prof = c.profiles()
    nshard = 100000 // nslice
    for r in (0, 1, nshard // 7, nshard // 3, nshard // 2, (2 * nshard) // 3, nshard - 2, nshard - 1):
        engine.init(c.tcfg, shard=(r, nshard))
        lo, hi = engine.local_range()
        ext = engine.lbl_extinction(prof)
        trm.free_memory()
        o = lbl_oracle.LblOracle(c.tcfg, wn_slice=(lo, hi))
        ref = o.extinction(prof, layers=list(layers))
        sel = list(layers)
        err = np.max(np.abs(ext[sel] - ref[sel]) / np.maximum(np.abs(ref[sel]), 1e-13 * ref.max()))
        worst = max(worst, err)
        print("wnosamp %4d  samples %6d..%6d (%.1f cm-1)  layers %s  max rel err %.2e" %
              (wnosamp, lo, hi, o.wn[0], sel, err), flush=True)
print("worst", worst)
sys.exit(0 if worst < 1e-7 else 1)
