"""The RT oracle against vectors of the REFERENCE's engine (exosports/transit).  None exist today: that engine is the
empty submodule of /root/reference/.gitmodules:8-10 and the reference holds no golden spectrum, so this test is
SKIPPED and RT parity is UNPINNED (oracle/rt_oracle.c header, DESIGN.md "Oracle").  `python tools/pin_reference.py <dir
of reference outputs> --write-golden` writes tests/golden/transit_ref_<name>/ -- a block of the reference's inputs, its
spectrum there and the combination of conventions that reproduced it -- and this test then holds the oracle (and, through
the GPU parity tests' oracle comparisons, the kernels) to them."""
import glob
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "transit_ref_*")))


def check_one(d, tol=1e-6):
    from oracle import rt_oracle as orc
    exp = np.load(os.path.join(d, "expected.npz"))
    combo = json.loads(str(exp["combination"]))
    cwd = os.getcwd()
    os.chdir(d)            # the block's cfg names its files relative to its own directory
    try:
        o = orc.OracleEngine("transit.cfg", integ=combo["integ"], cut=combo["cut"], cia_interp=combo["cia_interp"])
        atm = orc.read_atm(o.keys["atm"])
        prof = np.vstack([atm["temp"][None, :], atm["abund"].T])
        if "voigt" in combo:
            from oracle import lbl_oracle
            o.set_extra_extinction(lbl_oracle.LblOracle("transit.cfg", osamp_rule=combo["osamp_rule"], voigt=combo["voigt"]).extinction(prof))
        got = o.run(prof)
    finally:
        os.chdir(cwd)
    np.testing.assert_allclose(o.wn, exp["wn"], rtol=1e-9)
    np.testing.assert_allclose(got, exp["spectrum"], rtol=tol, atol=1e-12 * np.abs(exp["spectrum"]).max())


@pytest.mark.skipif(not CASES, reason="no vectors of the reference's engine exist (tools/pin_reference.py writes them): RT parity unpinned")
@pytest.mark.parametrize("d", CASES or ["none"])
def test_oracle_reproduces_the_reference_vectors(d):
    check_one(d)
