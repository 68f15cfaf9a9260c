"""Shapes outside the ahead-of-time set: seven / eight table molecules with two cross-section files
under the default spline (four slots), nine and more molecules, three cross-section files, ten and more ray angles used
to take the generic kernel, 3-4x slower on the few-walker launches.  They are instantiated from the same kernel templates at their first launch
(csrc/rtc.hpp, hiprtc), cached on disk, and held to the oracle like every other kernel; with BARTRT_RTC=0 the generic
kernel still serves them."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHILD = r"""
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
from bart_amd import engine, transit_module as trm
names = {}
for tcfg, pfile, out, nws, integ, cut in %(jobs)r:
    engine.init(tcfg)
    if integ is not None: trm.set_integ(integ)
    if cut is not None: trm.set_cut(cut)
    profs = np.load(pfile)
    res = []
    for n in nws:
        engine.walked_begin()
        res.append(engine.run_batch(profs[:n]))
        names["%%s|%%d" %% (out, n)] = engine.walked_end()[2]
    np.save(out, np.concatenate(res))
    trm.free_memory()
print("RES " + json.dumps({"names": names, "rtc": trm.get_rtc_stats()}))
"""


def run_child(jobs, env):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "jobs": jobs}], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RES ")][0][4:])


def test_shapes_outside_the_aot_set_are_instantiated_and_match_the_oracle(tmp_path):
    from bart_amd import synth
    from oracle import rt_oracle as orc
    from test_gpu_parity import many_molecules, walkers, RTOL
    cache = str(tmp_path / "cache")
    shapes = [("m7c4", dict(cia=2, **many_molecules(7)), None, None),            # two CS files, spline: four slots
              ("m8c4", dict(cia=2, **many_molecules(8)), None, None),
              ("m9c2", dict(cia=1, **many_molecules(9)), None, None),
              ("m4c6", dict(cia=3), None, None),                                 # three CS files
              ("a12", dict(raygrid=(0, 8, 16, 24, 32, 40, 48, 56, 64, 72, 78, 84)), None, None),
              # since round 6 NO ray grid other than the five-angle one, and no shape of seven or eight molecules, is
              # built ahead of time (the library went from 101 to 54 MB): the sizes that used to be are instantiated too
              ("a3", dict(raygrid=(0, 35, 70)), None, None),
              ("a7v", dict(raygrid=(0, 12, 25, 38, 51, 64, 77)), 0, "vertical"),
              ("m7c2", dict(cia=1, **many_molecules(7)), None, None),
              ("m9v0", dict(cia=1, **many_molecules(9)), 0, "vertical"),         # another rule, the other cut
              ("m7c4i2", dict(cia=2, **many_molecules(7)), 2, None)]
    jobs, cases = [], {}
    nws = (1, 3, 12, 70, 80)  # (5 columns per walker: 70 walkers = 350 columns, the adjacent-rows kernel's last range; 80 = 400, past it)
    for name, kw, integ, cut in shapes:
        c = synth.make_case(str(tmp_path / name), nlayers=61, nwave=300, tlow=400.0, thigh=3000.0, tempdelt=650.0, **kw)
        profs = walkers(c, 80, seed=31)
        np.save(os.path.join(c.dir, "p.npy"), profs)
        jobs.append((c.tcfg, os.path.join(c.dir, "p.npy"), os.path.join(c.dir, "s.npy"), nws, integ, cut))
        cases[name] = (c, integ, cut)
    rep = run_child(jobs, {"BARTRT_RTC_CACHE": cache})
    assert rep["rtc"]["available"] and rep["rtc"]["compiled"] >= 11 and rep["rtc"]["failed"] == 0, rep["rtc"]
    wide = ("m7c4", "m8c4", "m9c2", "m7c4i2")       # more than 20 loads per layer, single-wave kernel at 400 columns (m9v0: rule 0 + vertical cut keeps the layer-parallel kernel to 640)
    for key, kname in rep["names"].items():
        shape = os.path.basename(os.path.dirname(key.split("|")[0]))
        # (rule 2 has no adjacent-rows form: its wide batch is past the layer-parallel kernels' range at 350 columns already)
        if shape in wide and (key.endswith("|80") or (key.endswith("|70") and shape == "m7c4i2")):
            # the single-wave kernels spill on these shapes: their BATCHES never take one -- the generic kernel (measured
            # equal or faster), or a layer-parallel instantiation while the measured table still names one
            assert "generic" in kname or ("instantiated at run time" in kname and "simpson_slant" not in kname
                                          and "rt_eclipse_fast" not in kname), (key, kname)
        else:
            assert "generic" not in kname and "instantiated at run time" in kname, (key, kname)
    for name, (c, integ, cut) in cases.items():
        profs = np.load(os.path.join(c.dir, "p.npy"))
        o = orc.OracleEngine(c.tcfg, integ=integ, cut=cut)
        ref = o.run_batch(profs)
        got = np.load(os.path.join(c.dir, "s.npy"))
        want = np.concatenate([ref[:n] for n in nws])
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-12 * np.abs(want).max(), err_msg=name)
    # a second process finds the code objects on disk: nothing is compiled again
    rep2 = run_child(jobs[:2], {"BARTRT_RTC_CACHE": cache})
    assert rep2["rtc"]["compiled"] == 0 and rep2["rtc"]["from_disk"] >= 2, rep2["rtc"]
    # without a compiler the generic kernel serves the shape, to the same numbers
    first = np.load(jobs[0][2]).copy()
    rep3 = run_child(jobs[:1], {"BARTRT_RTC": "0", "BARTRT_RTC_CACHE": str(tmp_path / "none")})
    assert not rep3["rtc"]["available"] and all("generic" in k for k in rep3["names"].values()), rep3
    np.testing.assert_allclose(np.load(jobs[0][2]), first, rtol=1e-11, atol=1e-13 * np.abs(first).max())
