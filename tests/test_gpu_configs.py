"""GPU parity on the shapes BASELINE.json's configs name (other than the bench
line): the WASP-12b retrieval shape, the `direct` solution, and line-by-line
extinction feeding the transit geometry."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_bands(cfg, params_list, solution="eclipse"):
    """Independent chain: pinned host readers -> numpy T(p)/abundances -> C oracle
    RT -> numpy band integration."""
    from bart_amd import BARTfunc, hostio
    from oracle import pyhalf, rt_oracle as orc
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
    if solution == "eclipse":
        starfl, starwn, _, _ = hostio.readkurucz(wc.kurucz, ptargs[1], float(tep.getvalue("loggstar")[0]))
    for f in wc.filters:
        fwn, ftr = hostio.readfilter(f)
        a, b, ind = hostio.resample(o.wn, fwn, ftr, *( (starwn, starfl) if solution == "eclipse" else (fwn, ftr)))
        idx0.append(ind[0][0]); npts.append(len(ind[0])); nif.append(a); ist.append(b)
    out = []
    for par in params_list:
        prof, st = pyhalf.step_profiles(np.array(par), press, abund, species, wc.molfit, ptargs,
                                        wc.Tmin, wc.Tmax)
        assert st == 0
        out.append(pyhalf.bandflux(o.run(prof), o.wn, idx0, npts, np.concatenate(nif),
                                   np.concatenate(ist), rp / rstar, solution))
    return np.array(out)


def test_wasp12b_shape_batched(tmp_path):
    """examples/WASP-12b/BART.cfg shape: 100 layers x 2424 wavenumbers from 910
    cm-1, four opacity molecules, nine free parameters (5 T(p) + 4 abundances),
    four band-passes; ten walkers in one call (BASELINE config 4's batch)."""
    from bart_amd import BARTfunc, synthcfg
    mols = ("H2O", "CO", "CO2", "CH4")
    p0 = (-1.5, -0.8, -0.8, 0.5, 1.0, -0.3, 0.2, -0.5, 0.1)
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=2424, wnlow=910.0, opmol=mols,
                                          molfit=mols, params=p0, nfilters=4)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        assert (w.nwave, w.nfilters, w.nPT, w.nmolfit) == (2424, 4, 5, 4)
        rng = np.random.default_rng(4)
        pars = np.array(p0) + rng.normal(0, [0.1, 0.1, 0.1, 0.05, 0.01, 0.3, 0.3, 0.3, 0.3], (10, 9))
        pars[:, 3] = np.clip(pars[:, 3], 0, 1)
        band = w.step(pars)
        assert band.shape == (10, 4) and np.all(band > 0)
        ref = _oracle_bands(cfg, pars[:4])
        np.testing.assert_allclose(band[:4], ref, rtol=1e-9)
    finally:
        w.close()


def test_direct_solution(tmp_path):
    """solution = direct: no stellar division, plain filter averages of the
    emergent flux (BARTfunc.py:394-396, makecfg.py:101-102 maps it to eclipse)."""
    from bart_amd import BARTfunc, synthcfg
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=1500, solution="direct")
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        p = np.array([[-2.0, 0.0, 1.0, 0.0, 0.98, -0.5], [-2.2, 0.1, 0.7, 0.2, 0.95, 0.2]])
        band = w.step(p)
        ref = _oracle_bands(cfg, p, "direct")
        np.testing.assert_allclose(band, ref, rtol=1e-9)
        assert band.min() > 1e3            # fluxes, not planet-to-star ratios
    finally:
        w.close()


def test_line_by_line_transit_geometry(tmp_path):
    """On-the-fly Voigt extinction feeding the transmission geometry."""
    from bart_amd import engine, synth_lbl, transit_module as trm
    from oracle import lbl_oracle, rt_oracle as orc
    c = synth_lbl.make_lbl_case(str(tmp_path), nlines=1500, nwave=256, nlayers=20, cia=True,
                                extra_keys={"solution": "transit", "starrad": 1.145})
    engine.init(c.tcfg)
    try:
        prof = c.profiles()
        spec = trm.run_transit(prof.ravel(), trm.get_no_samples())
        o = orc.OracleEngine(c.tcfg)
        o.set_extra_extinction(lbl_oracle.LblOracle(c.tcfg).extinction(prof))
        np.testing.assert_allclose(spec, o.run(prof), rtol=1e-7)
        assert 0.01 < spec.min() < spec.max() < 0.03 and spec.std() > 0
    finally:
        trm.free_memory()


def test_every_whitelisted_transit_key_is_accepted(tmp_path):
    """A transit cfg as code/makecfg.py:36-104 writes it: every key of `known_args` that a
    BART cfg may carry, `shareOpacity` with no value, both comment styles, wavelength-based
    sampling.  The keys DESIGN.md section 7 lists as sampling / diagnostic controls of the CPU
    engine change nothing; the spectrum equals the bare configuration's bit for bit."""
    from bart_amd import engine, synth, transit_module as trm
    bare = synth.make_case(str(tmp_path / "bare"), nwave=300, nlayers=30)
    noeffect = {"radlow": "0", "radhigh": "100", "raddelt": "1", "radfct": "1e5", "allowq": "0.01",
                "tauiso": "0", "outtau": "0", "taulevel": "1", "modlevel": "1", "verb": "11",
                "savefiles": "no", "orbpars": "0.05 90 0 0 0 0", "orbparsfct": "1 1 1 1 1 1",
                "wnfct": "1.0", "shareOpacity": ""}
    full = synth.make_case(str(tmp_path / "full"), nwave=300, nlayers=30, extra_keys=noeffect)
    text = open(full.tcfg).read()
    for k in noeffect:
        assert ("\n" + k) in ("\n" + text), k
    with open(full.tcfg, "a") as f:
        f.write("; a comment line of the other style\n#another\n\n")
    spectra = []
    for c in (bare, full):
        engine.init(c.tcfg)
        try:
            spectra.append(trm.run_transit(c.profiles().ravel(), trm.get_no_samples()))
        finally:
            trm.free_memory()
    assert np.array_equal(spectra[0], spectra[1]) and spectra[0].min() > 0
    # wavelength-based sampling (examples/demo/transit_demo.cfg:17-24) gives the same grid
    lines = [l for l in open(bare.tcfg).read().split("\n") if l.split(" ")[0] not in ("wnlow", "wnhigh")]
    wn = bare.wn
    lines += ["wllow %.17g" % (1e4 / wn[-1]), "wlhigh %.17g" % (1e4 / wn[0]), "wlfct 1e-4"]
    wl_cfg = str(tmp_path / "wl.cfg")
    open(wl_cfg, "w").write("\n".join(lines) + "\n")
    engine.init(wl_cfg)
    try:
        assert trm.get_no_samples() == len(wn)
        np.testing.assert_allclose(trm.get_waveno_arr(len(wn)), wn, rtol=1e-12)
    finally:
        trm.free_memory()
