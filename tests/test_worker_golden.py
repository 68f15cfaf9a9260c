"""The reference's own per-step loop as the expected value: tests/golden/worker_golden.npz
holds what code/BARTfunc.py `main(comm)` -- imported from the reference and run unmodified by
tests/golden/make_worker_golden.py, with stand-ins for mpi4py, MC3's helpers and the RT
engine (the CPU oracle) -- built and returned on synthetic inputs: the profile arrays it
handed to run_transit, the setter calls, the band fluxes and -1 rows it gathered, the order
of its MPI calls.

CPU: the numpy restatement the device kernels are held to (oracle/pyhalf.py) reproduces
them (rows a3, a4, a8 of SURVEY.md 8 pinned to the reference's code path, not to a
restatement of it).  GPU: the product's worker returns the same band fluxes with the HIP
engine in the oracle's place."""
import json
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "worker_golden.npz"))
CASES = ["eclipse_ch4", "eclipse_4mol_cloud_ray", "transit_2mol", "direct_ch4", "eclipse_thorngren", "eclipse_polar",
         "pt_iso", "pt_madhu_inv", "pt_adiabatic", "pt_piette"]          # the loop's six temperature models
NPT = {"line": 5, "iso": 1, "madhu_noinv": 5, "madhu_inv": 6, "adiabatic": 3, "piette": 8}


def _rebuild(name, tmp_path):
    from bart_amd import synthcfg
    spec = json.loads(str(G[name + "_kw"]))
    kw = dict(spec["kw"])
    for k in ("opmol", "molfit", "params"):
        kw[k] = tuple(kw[k])
    case, cfg = synthcfg.make_worker_case(str(tmp_path), **kw)      # (kw may carry the TEP file's values)
    with open(cfg, "a") as f:
        for k, v in spec["extra"].items():
            f.write("%s = %s\n" % (k, v))
    if spec.get("pttype", "line") != "line":
        text = open(cfg).read().replace("PTtype = line", "PTtype = " + spec["pttype"])
        open(cfg, "w").write(text)
    sol = kw.get("solution", "eclipse")
    lay = dict(nPT=NPT[spec.get("pttype", "line")], nrad=int(sol == "transit"), ncloud=int("cloudtop" in spec["extra"]),
               nray=int("scattering" in spec["extra"]), solution=sol, pttype=spec.get("pttype", "line"))
    return case, cfg, lay


@pytest.mark.parametrize("name", CASES)
def test_restated_step_equals_the_reference_loop(name, tmp_path):
    from bart_amd import BARTfunc, hostio
    from oracle import pyhalf, rt_oracle as orc
    case, cfg, lay = _rebuild(name, tmp_path)
    wc = BARTfunc.WorkerConfig.from_cfg(cfg)
    tep = hostio.TepFile(wc.tep_name)
    rstar = float(tep.getvalue("Rs")[0]) * hostio.Rsun
    rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
    mp = float(tep.getvalue("Mp")[0]) * hostio.Mjup
    ptargs = [rstar, float(tep.getvalue("Ts")[0]), wc.tint, float(tep.getvalue("a")[0]) * hostio.AU,
              100.0 * hostio.G_NEWTON * mp / rp ** 2]
    species, press, _, abund = hostio.readatm(wc.atmfile)
    o = orc.OracleEngine(wc.tconfig)
    idx0, npts, nif, ist = [], [], [], []
    if lay["solution"] != "direct":
        starfl, starwn, _, _ = hostio.readkurucz(wc.kurucz, ptargs[1], float(tep.getvalue("loggstar")[0]))
    for f in wc.filters:
        fwn, ftr = hostio.readfilter(f)
        a, b, ind = hostio.resample(o.wn, fwn, ftr, *((starwn, starfl) if lay["solution"] != "direct" else (fwn, ftr)))
        idx0.append(ind[0][0]); npts.append(len(ind[0])); nif.append(a); ist.append(b)
    pars, band, acc = G[name + "_params"], G[name + "_band"], list(G[name + "_accepted"])
    nextra = lay["nrad"] + lay["ncloud"] + lay["nray"]
    calls, j = [], 0
    n = lay["nPT"]
    assert 0 < len(acc) < len(pars) and wc.PTtype == lay["pttype"]
    for i, par in enumerate(pars):
        core = np.concatenate([par[:n], par[n + nextra:]])
        prof, st = pyhalf.step_profiles(core, press, abund, species, wc.molfit, ptargs if lay["pttype"] == "line" else (),
                                        wc.Tmin, wc.Tmax, pttype=lay["pttype"], t_int_type=wc.tint_type)
        rejected = bool(np.all(band[i] == -1.0))
        assert (st != 0) == rejected == (i not in acc), (i, st)
        if rejected:
            continue
        # the profile array the reference handed to run_transit: abundances to the bit
        # (same operations), temperatures to the accuracy pyhalf's T(p) is pinned at
        ref_prof = G[name + "_profiles"][j].reshape(prof.shape)
        np.testing.assert_array_equal(prof[1:], ref_prof[1:])
        np.testing.assert_allclose(prof[0], ref_prof[0], rtol=1e-13)
        # setters in the reference's order (BARTfunc.py:350-360), then the engine, then the bands
        if lay["nrad"]:
            o.set_radius(par[n]); calls.append(("set_radius", j, par[n], 0.0))
        if lay["ncloud"]:
            o.set_cloudtop(par[n + lay["nrad"]]); calls.append(("set_cloudtop", j, par[n + lay["nrad"]], 0.0))
        if lay["nray"] and "polar" in wc.scattering:
            o.set_scattering(2, 0.0); calls.append(("set_scattering", j, 2.0, 0.0))
        elif lay["nray"]:
            o.set_scattering(1, par[n + lay["nrad"] + lay["ncloud"]])
            calls.append(("set_scattering", j, 1.0, par[n + lay["nrad"] + lay["ncloud"]]))
        spec = o.run(prof)
        np.testing.assert_allclose(spec, G[name + "_spectra"][j], rtol=1e-11)
        sol = "eclipse" if lay["solution"] == "eclipse" else "direct"    # transit bands are plain filter means too
        got = pyhalf.bandflux(spec, o.wn, idx0, npts, np.concatenate(nif),
                              np.concatenate(ist) if ist[0] is not None else None, rp / rstar, sol)
        np.testing.assert_allclose(got, band[i], rtol=1e-11)
        j += 1
    assert j == len(acc)
    names = [c for c in G[name + "_call_names"] if c != "free_memory"]
    assert names == [c[0] for c in calls]
    if calls:
        np.testing.assert_array_equal(G[name + "_calls"], np.array([[c[1], c[2], c[3]] for c in calls]))
    # MC3's protocol as the reference speaks it: one broadcast, scatter / gather per step,
    # the scatter that brings the end flag, disconnect
    assert list(G[name + "_log"]) == ["bcast"] + ["scatter", "gather"] * len(pars) + ["scatter", "disconnect"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_worker_equals_the_reference_loop(name, tmp_path):
    """The product's worker (HIP engine) on the same inputs and parameter vectors: the band
    fluxes the reference's loop gathered -- 1e-9 on the accepted steps (the spectra inside the
    golden run are the CPU oracle's), the -1 rows where it rejected -- one walker at a time as
    the reference runs, and all steps as one batch; and the MPI calls in the same order."""
    from bart_amd import BARTfunc
    from test_worker import FakeIntercomm
    case, cfg, lay = _rebuild(name, tmp_path)
    pars, band = G[name + "_params"], G[name + "_band"]
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        batch = w.step(pars)
        single = np.array([w.step(p)[0] for p in pars])
    finally:
        w.close()
    rej = np.all(band == -1.0, axis=1)
    for got in (batch, single):
        assert np.array_equal(got[rej], band[rej])
        np.testing.assert_allclose(got[~rej], band[~rej], rtol=1e-9)
    comm = FakeIntercomm(list(pars))
    BARTfunc.main(comm, ["BARTfunc.py", "-c", cfg])
    got = np.array(comm.received)
    assert np.array_equal(got[rej], band[rej])
    np.testing.assert_allclose(got[~rej], band[~rej], rtol=1e-9)
    log = [x for x in comm.log if x != "barrier"]
    assert log == [x for x in G[name + "_log"] if x != "disconnect"] and comm.disconnected


def _makecfg_case(tmp_path):
    from bart_amd import synthcfg
    kw = json.loads(str(G["makecfg_case_kw"]))
    for k in ("opmol", "molfit"):
        kw[k] = tuple(kw[k])
    case, _ = synthcfg.make_worker_case(str(tmp_path), **kw)
    tcfg = os.path.join(str(tmp_path), "made_transit.cfg")
    open(tcfg, "w").write(str(G["makecfg_transit_text"]).replace("@DIR@", str(tmp_path)))
    return case, tcfg


def test_transit_cfg_written_by_makecfg_is_read(tmp_path):
    """The text code/makecfg.py `makeTransit` wrote for a BART cfg (two csfile lines joined
    by a comma, one `linedb` line per TLI, refradius / gsurf from the TEP file, `direct`
    mapped to eclipse, a bare `shareOpacity`, none of the sampler's keys), through the
    oracle's reader; the engine's reader sees the same file in the GPU test below."""
    from oracle import rt_oracle as orc
    case, tcfg = _makecfg_case(tmp_path)
    k = orc.read_tcfg(tcfg)
    d = str(tmp_path)
    assert k["csfile"] == "%s/CIA_H2H2.dat,%s/CIA_H2He.dat" % (d, d)
    assert k["linedb"] == "%s/a.tli,%s/b.tli" % (d, d)
    assert k["solution"] == "eclipse" and k["refradius"] == "96514.20" and k["gsurf"] == "897.7"
    assert "shareOpacity" in k and "burnin" not in k and "walk" not in k and "params" not in k
    o = orc.OracleEngine(tcfg)
    assert len(o.wn) == 200 and o.wn[0] == 2500.0 and o.wn[-1] == 2699.0


@pytest.mark.gpu
def test_engine_runs_the_cfg_makecfg_wrote(tmp_path):
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    case, tcfg = _makecfg_case(tmp_path)
    prof = case.profiles()
    engine.init(tcfg)
    try:
        n = trm.get_no_samples()
        assert n == 200
        spec = trm.run_transit(prof.ravel(), n)
    finally:
        trm.free_memory()
    ref = orc.OracleEngine(tcfg).run(prof)
    np.testing.assert_allclose(spec, ref, rtol=1e-10)


CARRY = "eclipse_madhu_valueerror"


def test_valueerror_steps_of_the_reference_loop(tmp_path):
    """BARTfunc.py:318-330 as the reference ran it: PT_NoInversion raises ValueError on four of
    the eight steps, the loop goes on with the temperature array the previous step left (zeros
    on the very first step -> rejected; a too-hot rejected profile carried into the next
    failing step -> rejected again; otherwise the old temperatures with the NEW abundances).
    The restated loop reproduces its profile arrays and band fluxes."""
    from bart_amd import BARTfunc, hostio
    from oracle import pyhalf, rt_oracle as orc
    case, cfg, lay = _rebuild(CARRY, tmp_path)
    wc = BARTfunc.WorkerConfig.from_cfg(cfg)
    assert wc.PTtype == "madhu_noinv"
    tep = hostio.TepFile(wc.tep_name)
    rstar = float(tep.getvalue("Rs")[0]) * hostio.Rsun
    rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
    species, press, _, abund = hostio.readatm(wc.atmfile)
    o = orc.OracleEngine(wc.tconfig)
    starfl, starwn, _, _ = hostio.readkurucz(wc.kurucz, float(tep.getvalue("Ts")[0]),
                                             float(tep.getvalue("loggstar")[0]))
    idx0, npts, nif, ist = [], [], [], []
    for f in wc.filters:
        a, b, ind = hostio.resample(o.wn, *hostio.readfilter(f), starwn, starfl)
        idx0.append(ind[0][0]); npts.append(len(ind[0])); nif.append(a); ist.append(b)
    sp = list(species)
    ih2, ihe, ich4 = sp.index("H2"), sp.index("He"), sp.index("CH4")
    ratio = abund[:, ih2] / abund[:, ihe]
    imetals = [i for i, x in enumerate(sp) if x not in ("He", "H2", "H-", "e-")]
    pars, band = G[CARRY + "_params"], G[CARRY + "_band"]
    assert [bool(np.all(b == -1)) for b in band] == [True, False, False, False, False, True, True, False]
    tprofile, j = np.zeros(len(press)), 0
    for i, par in enumerate(pars):
        try:
            tprofile[:] = pyhalf.pt_noinversion(np.asarray(press)[::-1], *par[:5])[::-1]
        except ValueError:
            pass
        if np.any(tprofile < wc.Tmin) or np.any(tprofile > wc.Tmax):
            assert np.all(band[i] == -1)
            continue
        prof = np.zeros((len(sp) + 1, len(press)))
        prof[0], prof[1:] = tprofile, abund.T
        prof[1 + ich4] = abund[:, ich4] * 10.0 ** par[5]
        q = 1.0 - prof[1:][imetals].sum(axis=0)
        assert not np.any(q < 0)
        prof[1 + ih2], prof[1 + ihe] = ratio * q / (1 + ratio), q / (1 + ratio)
        ref_prof = G[CARRY + "_profiles"][j].reshape(prof.shape)
        np.testing.assert_array_equal(prof[1:], ref_prof[1:])
        np.testing.assert_allclose(prof[0], ref_prof[0], rtol=1e-13)
        got = pyhalf.bandflux(o.run(prof), o.wn, idx0, npts, np.concatenate(nif), np.concatenate(ist), rp / rstar)
        np.testing.assert_allclose(got, band[i], rtol=1e-11)
        j += 1
    assert j == len(G[CARRY + "_accepted"]) == 5
    assert not np.allclose(band[1], band[2])               # the carried profile met a new abundance


@pytest.mark.gpu
def test_worker_carries_the_profile_like_the_reference_loop(tmp_path, monkeypatch):
    """The product's one-chain worker with BARTRT_CARRY_PROFILE=1 (bartrt_step_set_carry): the
    -1 / band-flux sequence the reference's loop returned on the ValueError case; without the
    switch those steps are rejected."""
    from bart_amd import BARTfunc
    from test_worker import FakeIntercomm
    case, cfg, lay = _rebuild(CARRY, tmp_path)
    pars, band = G[CARRY + "_params"], G[CARRY + "_band"]
    rej = np.all(band == -1.0, axis=1)
    monkeypatch.setenv("BARTRT_CARRY_PROFILE", "1")
    comm = FakeIntercomm(list(pars))
    BARTfunc.main(comm, ["-c", cfg])
    got = np.array(comm.received)
    assert np.array_equal(got[rej], band[rej])
    np.testing.assert_allclose(got[~rej], band[~rej], rtol=1e-9)
    monkeypatch.delenv("BARTRT_CARRY_PROFILE")
    comm = FakeIntercomm(list(pars))
    BARTfunc.main(comm, ["-c", cfg])
    got = np.array(comm.received)
    raises = [0, 2, 4, 6]                                   # the steps whose model raises
    assert np.all(got[raises] == -1.0)
    np.testing.assert_allclose(got[[1, 3, 7]], band[[1, 3, 7]], rtol=1e-9)


EBAL = "eclipse_ebalance"


def _ebalance_expected(tmp_path):
    """The restated step with the energy-balance branch of BARTfunc.py:365-382."""
    from bart_amd import BARTfunc, hostio
    from oracle import pyhalf, rt_oracle as orc
    case, cfg, lay = _rebuild(EBAL, tmp_path)
    wc = BARTfunc.WorkerConfig.from_cfg(cfg)
    assert wc.ebalance and wc.tint == 800.0
    tep = hostio.TepFile(wc.tep_name)
    tstar = float(tep.getvalue("Ts")[0])
    rstar = float(tep.getvalue("Rs")[0]) * hostio.Rsun
    sma = float(tep.getvalue("a")[0]) * hostio.AU
    rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
    mp = float(tep.getvalue("Mp")[0]) * hostio.Mjup
    ptargs = [rstar, tstar, wc.tint, sma, 100.0 * hostio.G_NEWTON * mp / rp ** 2]
    species, press, _, abund = hostio.readatm(wc.atmfile)
    o = orc.OracleEngine(wc.tconfig)
    starfl, starwn, _, _ = hostio.readkurucz(wc.kurucz, tstar, float(tep.getvalue("loggstar")[0]))
    idx0, npts, nif, ist = [], [], [], []
    for f in wc.filters:
        a, b, ind = hostio.resample(o.wn, *hostio.readfilter(f), starwn, starfl)
        idx0.append(ind[0][0]); npts.append(len(ind[0])); nif.append(a); ist.append(b)
    e_in = hostio.sig * tstar ** 4 * rstar ** 2 * np.pi * rp ** 2 / sma ** 2 * 1e7
    want, margin = [], []
    for i, par in enumerate(G[EBAL + "_params"]):
        prof, st = pyhalf.step_profiles(par, press, abund, species, wc.molfit, ptargs, wc.Tmin, wc.Tmax)
        assert st == 0                                       # every step reaches the engine
        ref_prof = G[EBAL + "_profiles"][i].reshape(prof.shape)
        np.testing.assert_array_equal(prof[1:], ref_prof[1:])
        np.testing.assert_allclose(prof[0], ref_prof[0], rtol=1e-13)
        spec = o.run(prof)
        e_out = pyhalf.energy_out(spec, o.wn, rp)
        margin.append(abs(e_out / e_in - 1))
        want.append(-np.ones(len(idx0)) if e_out > e_in else
                    pyhalf.bandflux(spec, o.wn, idx0, npts, np.concatenate(nif), np.concatenate(ist), rp / rstar))
    return cfg, np.array(want), min(margin)


def test_energy_balance_steps_of_the_reference_loop(tmp_path):
    """Some of the eight steps emit more in the modelled band than the planet receives and
    are answered with -1 AFTER the engine ran (the loop made eight run_transit calls)."""
    cfg, want, margin = _ebalance_expected(tmp_path)
    band = G[EBAL + "_band"]
    rej = np.all(band == -1.0, axis=1)
    assert 2 <= rej.sum() <= 6 and len(G[EBAL + "_profiles"]) == 8 and margin > 1e-6
    assert np.array_equal(np.all(want == -1.0, axis=1), rej)
    np.testing.assert_allclose(want[~rej], band[~rej], rtol=1e-11)


@pytest.mark.gpu
def test_worker_energy_balance_like_the_reference_loop(tmp_path):
    from bart_amd import BARTfunc
    from test_worker import FakeIntercomm
    case, cfg, lay = _rebuild(EBAL, tmp_path)
    pars, band = G[EBAL + "_params"], G[EBAL + "_band"]
    rej = np.all(band == -1.0, axis=1)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        batch = w.step(pars)
    finally:
        w.close()
    comm = FakeIntercomm(list(pars))
    BARTfunc.main(comm, ["-c", cfg])
    for got in (batch, np.array(comm.received)):
        assert np.array_equal(got[rej], band[rej])
        np.testing.assert_allclose(got[~rej], band[~rej], rtol=1e-9)
