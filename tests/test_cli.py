"""The standalone `transit` executable and its output files,
read back the way the reference's post-processing reads them."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bart_amd", "transit")


def read_spectrum(path):
    """Parsing rule of code/readtransit.py:38-62: skip one header line, first
    column wavelength (um), last column the value."""
    rows = [ln.split() for ln in open(path).read().strip().split("\n")[1:]]
    return 1e4 / np.array([float(r[0]) for r in rows]), np.array([float(r[-1]) for r in rows])


def read_tau_dat(path, nlayers):
    """Parsing rule of code/cf.py:68-94."""
    lines = open(path).readlines()
    while lines[0].startswith("#") or not lines[0].strip():
        lines.pop(0)
    tau_lines, wn_lines = lines[1:-1:3], lines[0:-1:3]
    tau = np.zeros((len(tau_lines), nlayers))
    wns = np.zeros(len(wn_lines))
    for i in range(len(tau_lines)):
        tau[i] = tau_lines[i].split()
        wns[i] = float(wn_lines[i].split()[1])
    return tau.T, wns


def test_cli_usage_without_gpu():
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([CLI, "-c", "/nonexistent.cfg"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_cli_outputs(tmp_path):
    from bart_amd import synth
    from oracle import rt_oracle as orc
    d = str(tmp_path)
    c = synth.make_case(d, nwave=200, nlayers=40, extra_keys={
        "outspec": os.path.join(d, "spec.dat"), "outtoomuch": os.path.join(d, "toom.dat"),
        "outintens": os.path.join(d, "intens.dat"), "outsample": os.path.join(d, "sample.dat"),
        "savefiles": "yes"})
    r = subprocess.run([CLI, "-c", c.tcfg], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    o = orc.OracleEngine(c.tcfg)
    ref, rtau, rlast = o.run(c.profiles(), want_tau=True)
    wn, spec = read_spectrum(os.path.join(d, "spec.dat"))
    np.testing.assert_allclose(wn, o.wn, rtol=1e-8)
    np.testing.assert_allclose(spec, ref, rtol=2e-9)              # 9 significant digits on file
    tau, wns = read_tau_dat(os.path.join(d, "tau.dat"), 40)
    assert tau.shape == (40, 200)
    np.testing.assert_allclose(wns, o.wn, rtol=1e-8)
    np.testing.assert_allclose(tau.T, rtau, rtol=2e-9, atol=1e-300)
    _, inten = read_spectrum(os.path.join(d, "intens.dat"))       # last column = 80 deg ray
    np.testing.assert_allclose(inten, o.intensity(c.profiles())[-1], rtol=2e-9)
    wl, rtm = np.loadtxt(os.path.join(d, "toom.dat"), unpack=True)
    _, rad = o.extinction(c.profiles())
    reached = rtau[np.arange(200), rlast] > 10.0
    np.testing.assert_allclose(rtm[reached], rad[::-1][rlast[reached]] / 1e5, rtol=1e-8)
    assert np.all(rtm[~reached] == 0)
    assert "radius sampling" in open(os.path.join(d, "sample.dat")).read()


@pytest.mark.gpu
def test_cli_just_opacity(tmp_path):
    from bart_amd import synth_lbl
    c = synth_lbl.make_lbl_case(str(tmp_path), nlines=200, nwave=60, nlayers=5, with_table=True,
                                tlow=800.0, thigh=1600.0, tempdelt=800.0)
    r = subprocess.run([CLI, "-c", c.tcfg, "--justOpacity"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(c.keys["opacityfile"]) > 5 * 2 * 2 * 60 * 8


@pytest.mark.gpu
def test_c_host_on_the_c_abi(tmp_path):
    """examples/c_host.c, compiled as C99 against include/bartrt.h: init, sizes, the
    atmosphere file's profile through run_transit, a batch of three -- against the
    oracle and the Python shim."""
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    c = synth.make_case(str(tmp_path / "case"), nwave=500, nlayers=60)
    exe = str(tmp_path / "c_host")
    libdir = os.path.join(ROOT, "bart_amd")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_host.c"), "-L" + libdir, "-lbartrt",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = str(tmp_path / "spec.txt")
    r = subprocess.run([exe, c.tcfg, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "500 samples, ok flags 1 1 1" in r.stdout
    got = np.loadtxt(out)
    o = orc.OracleEngine(c.tcfg)
    assert np.array_equal(got[:, 0], o.wn)
    prof = c.profiles()
    np.testing.assert_allclose(got[:, 1], o.run(prof.ravel()), rtol=1e-10)
    for col, f in ((3, 1.05), (4, 0.95)):
        p = prof.copy()
        p[0] *= f
        np.testing.assert_allclose(got[:, col], o.run(p.ravel()), rtol=1e-10)
    np.testing.assert_allclose(got[:, 2], got[:, 1], rtol=1e-12)      # first row of the batch
    engine.init(c.tcfg)
    try:
        assert np.array_equal(trm.run_transit(prof.ravel(), 500), got[:, 1])
    finally:
        trm.free_memory()


REF_CODE = "/root/reference/code"


@pytest.mark.skipif(not os.path.isdir(REF_CODE), reason="reference code not on this box")
def test_reference_readers_on_product_files():
    """The reference's OWN readers -- code/readtransit.py `readspectrum` (bestFit.py:578-586
    reads the best-fit spectrum with it) and code/cf.py `readTauDat` (the contribution
    functions' input) -- imported and run on files the product's `transit` executable wrote on
    an MI355X (tests/golden/cli_outputs/, tools/make_cli_fixture.py), against the same
    quantities taken from the library's API in the same run (expected.npz).  matplotlib
    (imported by both modules for their plots) is stubbed where it is not installed."""
    import sys
    import types
    G = os.path.join(ROOT, "tests", "golden", "cli_outputs")
    exp = np.load(os.path.join(G, "expected.npz"))
    added = [n for n in ("float", "int") if not hasattr(np, n)]
    for n in added:                                      # cf.py:89 uses np.float (gone from numpy >= 1.24)
        setattr(np, n, {"float": float, "int": int}[n])
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        mpl = types.ModuleType("matplotlib")
        mpl.use = lambda *a, **k: None
        for sub in ("pyplot", "gridspec"):
            m = types.ModuleType("matplotlib." + sub)
            setattr(mpl, sub, m)
            sys.modules["matplotlib." + sub] = m
        sys.modules["matplotlib"] = mpl
    sys.path.insert(0, REF_CODE)
    nobytecode = sys.dont_write_bytecode
    sys.dont_write_bytecode = True                       # the reference tree is read-only: no __pycache__ there
    try:
        import cf
        import readtransit
        wn, spec = readtransit.readspectrum(os.path.join(G, "spec.dat"), wn=True)
        tau, wns = cf.readTauDat(os.path.join(G, "tau.dat"), int(exp["nlayers"]))
        _, inten = readtransit.readspectrum(os.path.join(G, "intens.dat"), wn=True)
    finally:
        sys.dont_write_bytecode = nobytecode
        sys.path.remove(REF_CODE)
        for n in added:
            delattr(np, n)
    np.testing.assert_allclose(wn, exp["wn"], rtol=1e-8)
    np.testing.assert_allclose(spec, exp["spectrum"], rtol=2e-9)          # 9 significant digits on file
    assert tau.shape == (int(exp["nlayers"]), len(exp["wn"]))
    np.testing.assert_allclose(wns, exp["wn"], rtol=1e-8)
    np.testing.assert_allclose(tau.T, exp["tau"], rtol=2e-9, atol=1e-300)
    assert inten.shape == spec.shape and np.all(inten > 0)
    # and the restated parsing rules the GPU tests use agree with the reference's readers
    wn2, spec2 = read_spectrum(os.path.join(G, "spec.dat"))
    tau2, wns2 = read_tau_dat(os.path.join(G, "tau.dat"), int(exp["nlayers"]))
    assert np.array_equal(wn2, wn) and np.array_equal(spec2, spec)
    assert np.array_equal(tau2, tau) and np.array_equal(wns2, wns)
