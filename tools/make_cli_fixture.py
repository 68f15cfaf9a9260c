"""Product output files for tests/golden/cli_outputs/ (GPU box): the standalone `transit`
executable on a small seeded case -- outspec, outintens, outtoomuch, tau.dat -- next to the
same quantities taken from the library's API (expected.npz).  The committed copies are
read by the REFERENCE's own readers (code/readtransit.py readspectrum, code/cf.py
readTauDat) in tests/test_cli.py::test_reference_readers_on_product_files.
    python tools/make_cli_fixture.py gpurun_out/cli_fixture"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bart_amd import engine, synth, transit_module as trm  # noqa: E402

out = os.path.abspath(sys.argv[1])
os.makedirs(out, exist_ok=True)
d = os.path.join(out, "_inputs")
c = synth.make_case(d, nwave=48, nlayers=12, extra_keys={
    "outspec": os.path.join(out, "spec.dat"), "outtoomuch": os.path.join(out, "toom.dat"),
    "outintens": os.path.join(out, "intens.dat"), "savefiles": "yes"})
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bart_amd", "transit")
subprocess.check_call([cli, "-c", c.tcfg], cwd=out)
engine.init(c.tcfg)
prof = c.profiles().ravel()
n = trm.get_no_samples()
spec = trm.run_transit(prof, n)
tau, last = engine.get_tau()
np.savez(os.path.join(out, "expected.npz"), wn=trm.get_waveno_arr(n), spectrum=spec, tau=tau, last=last,
         nlayers=12)
trm.free_memory()
import shutil  # noqa: E402
shutil.rmtree(d)
print(sorted(os.listdir(out)))
