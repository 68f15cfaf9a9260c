"""`transit --justOpacity` equivalent: time to build the opacity grid
o[100 layers][27 temperatures][4 molecules][1e4 wavenumbers] (864 MB) from a
1e6-line synthetic list on the GPU (done inside bartrt_init when the configured
opacity file does not exist yet)."""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bart_amd import engine, synth_lbl, transit_module as trm  # noqa: E402

d = os.path.join(tempfile.gettempdir(), "bartrt_opgen")
shutil.rmtree(d, ignore_errors=True)
mols = ("H2O", "CO", "CO2", "CH4")
case = synth_lbl.make_lbl_case(d, molecules=mols, nlines=250000, nwave=10000, wnlow=1000.0, wndelt=1.0,
                               nlayers=100, cia=True, with_table=True)
t0 = time.perf_counter()
engine.init(case.tcfg)
dt = time.perf_counter() - t0
size = os.path.getsize(os.path.join(d, "opacity_from_lines.dat"))
trm.free_memory()
t0 = time.perf_counter()
engine.init(case.tcfg)              # second init: reads the file it wrote
dt2 = time.perf_counter() - t0
trm.free_memory()
print(json.dumps({"workload": "opacity grid from 1e6 lines: 100 layers x 27 T x 4 molecules x 1e4 wavenumbers",
                  "init_with_generation_s": dt, "init_reading_the_grid_s": dt2, "grid_bytes": size}))
shutil.rmtree(d, ignore_errors=True)
