"""tools/pin_reference.py, exercised with the ORACLE playing the reference's engine under a hidden
combination of the unverified conventions: the tool must validate the inputs through the product's readers, find the
hidden combination by its error, and write golden vectors tests/test_reference_pin.py can hold the oracle to.  This is
readiness -- it pins nothing: the real engine (exosports/transit) is the empty submodule of the reference checkout."""
import json
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def write_transit_spectrum(path, wn, flux):
    """The layout code/readtransit.py:23-64 reads: a comment line, then wavelength (um) ... flux rows."""
    with open(path, "w") as f:
        f.write("#wavelength [um]    flux [erg/s/cm]\n")
        for w, v in zip(wn[::-1], flux[::-1]):
            f.write("%.17e    %.17e\n" % (1e4 / w, v))


def write_tau_dat(path, wn, tau):
    """code/cf.py:68-94: per sample a `wn` line, the optical depths top -> bottom, one more line."""
    with open(path, "w") as f:
        f.write("# optical depth per layer\n\n")
        for i, w in enumerate(wn):
            f.write("wn %.10f\n" % w)
            f.write(" ".join("%.17e" % t for t in tau[i]) + "\n")
            f.write("last\n")
        f.write("end\n")


@pytest.fixture()
def refdir(tmp_path):
    """What a run of the reference would leave behind -- made by the oracle under integ 2 / cut vertical / cia linear,
    with a cloud top in the cfg."""
    from bart_amd import synth
    from oracle import rt_oracle as orc
    d = tmp_path / "ref"
    case = synth.make_case(str(d), nlayers=30, nwave=240, cia=2, extra_keys={"cloudtop": -0.5, "outspec": "hidden.spec"})
    os.rename(case.tcfg, str(d / "hidden.cfg"))
    o = orc.OracleEngine(str(d / "hidden.cfg"), integ=2, cut="vertical", cia_interp="linear")
    prof = case.profiles()
    spec, tau, last = o.run(prof, want_tau=True)
    write_transit_spectrum(str(d / "hidden.spec"), o.wn, spec)
    idx = np.arange(0, 240, 24)
    write_tau_dat(str(d / "tau.dat"), o.wn[idx], tau[idx])
    return str(d)


def test_the_hidden_combination_is_found_and_golden_vectors_written(refdir, tmp_path, monkeypatch):
    import pin_reference
    out = tmp_path / "report.json"
    golden = os.path.join(ROOT, "tests", "golden", "transit_ref_hidden")
    try:
        rc = pin_reference.main([refdir, "--write-golden", "--samples", "48", "--json", str(out)])
        assert rc == 0
        rep, = json.load(open(out))
        assert [v for _, _, v in rep["inputs"]] == ["ok"] * len(rep["inputs"]) and len(rep["inputs"]) >= 6
        w = rep["winner"]
        assert (w["integ"], w["cut"], w["cia_interp"], w["C11_cloud_scattering_keys"]) == (2, "vertical", "linear", "honoured")
        assert w["max_rel_err"] < 1e-12 and w["tau_max_rel_err"] < 1e-12
        # every other combination is told apart by far more than the contract's 1e-6
        assert rep["runner_up_max_rel_err"] > 1e-5
        assert len(rep["combinations"]) == 3 * 2 * 2 * 2
        # the golden block: small, self-contained, and the oracle reproduces it
        assert os.path.isdir(golden)
        size = sum(os.path.getsize(os.path.join(golden, f)) for f in os.listdir(golden))
        assert size < 4 << 20
        import test_reference_pin
        test_reference_pin.check_one(golden)
    finally:
        shutil.rmtree(golden, ignore_errors=True)


def test_a_corrupt_input_is_named(refdir):
    import pin_reference
    keys = pin_reference.read_keys(os.path.join(refdir, "hidden.cfg"))
    atm = keys["atm"]
    txt = open(atm).read().replace("#TEADATA", "#TEADATA_", 1)
    open(atm, "w").write(txt)
    exe = os.path.join(refdir, "validate")
    import subprocess
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tools", "fuzz_readers.cpp"),
                           os.path.join(ROOT, "bart_amd", "csrc", "io.cpp"), "-o", exe])
    rep = pin_reference.validate_inputs(os.path.join(refdir, "hidden.cfg"), exe)
    assert rep[-1][0] == "atm" and rep[-1][2].startswith("IoError")


def test_a_reference_on_another_grid_is_reported_not_compared(refdir, tmp_path):
    import pin_reference
    spec = os.path.join(refdir, "hidden.spec")
    lines = open(spec).read().split("\n")
    open(spec, "w").write("\n".join(lines[:-20]) + "\n")
    rc = pin_reference.main([refdir, "--json", str(tmp_path / "r.json")])
    assert rc == 1
    rep, = json.load(open(tmp_path / "r.json"))
    assert "grid" in rep["error"]
