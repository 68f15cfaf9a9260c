"""HIP path vs the CPU oracle through the C ABI (ctypes), same seeded inputs.

Tolerance: north_star states 1e-6 relative; the fp64 kernels are held to 1e-10
here (differences come only from FMA contraction and libm vs ocml exp)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def walkers(case, n, seed=3):
    rng = np.random.default_rng(seed)
    L = len(case.press_bar)
    out = []
    for _ in range(n):
        t = case.temp0 + rng.uniform(-300, 600) + 150 * np.sin(
            np.linspace(0, rng.uniform(1, 6), L) + rng.uniform(0, 6))
        t = np.clip(t, 410.0, 2990.0)
        ab = case.abund0.copy()
        for s in range(2, ab.shape[1]):
            ab[:, s] *= 10 ** rng.uniform(-2, 1)
        q = 1 - ab[:, 2:].sum(1)
        ab[:, 1] = 0.85 / 0.15 * q / (1 + 0.85 / 0.15)
        ab[:, 0] = q / (1 + 0.85 / 0.15)
        out.append(case.profiles(t, ab).ravel())
    return np.array(out)


def test_single_walker_matches_oracle(small_case):
    from bart_amd import transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    trm.transit_init(3, ["transit", "-c", c.tcfg])
    try:
        n = trm.get_no_samples()
        wn = trm.get_waveno_arr(n)
        o = orc.OracleEngine(c.tcfg)
        assert n == len(o.wn) and np.array_equal(wn, o.wn)
        prof = c.profiles().ravel()
        spec = trm.run_transit(prof, n)
        ref = o.run(prof)
        assert np.all(np.isfinite(spec)) and spec.min() > 0
        np.testing.assert_allclose(spec, ref, rtol=RTOL)
    finally:
        trm.free_memory()


def test_tau_and_last_match_oracle(small_case):
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    engine.init(c.tcfg)
    try:
        prof = walkers(c, 1, seed=11)[0]
        trm.run_transit(prof, trm.get_no_samples())
        tau, last = engine.get_tau()
        _, rtau, rlast = orc.OracleEngine(c.tcfg).run(prof, want_tau=True)
        assert np.array_equal(last, rlast)
        np.testing.assert_allclose(tau, rtau, rtol=RTOL, atol=1e-300)
    finally:
        trm.free_memory()


def test_tau_after_a_batch_needs_the_walker_named(small_case):
    """bartrt_get_tau / _get_intensity serve the latest single-profile host call; after a batch they
    refuse (no stale or silently-first-walker data) and bartrt_get_tau_of /
    _get_intensity_of return the named walker's arrays."""
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    engine.init(c.tcfg)
    try:
        o = orc.OracleEngine(c.tcfg)
        profs = walkers(c, 4, seed=21)
        n = trm.get_no_samples()
        engine.run_batch(profs)
        with pytest.raises(trm.TransitError, match="batch"):
            engine.get_tau()
        inten = np.zeros((5, n))
        assert trm.lib().bartrt_get_intensity(trm._ptr(inten), 5, n) < 0
        for w in (0, 3):
            tau, last = engine.get_tau(walker=w)
            _, rtau, rlast = o.run(profs[w], want_tau=True)
            assert np.array_equal(last, rlast)
            np.testing.assert_allclose(tau, rtau, rtol=RTOL, atol=1e-300)
            trm.check(trm.lib().bartrt_get_intensity_of(w, trm._ptr(inten), 5, n))
            np.testing.assert_allclose(inten, o.intensity(profs[w]), rtol=RTOL)
        with pytest.raises(trm.TransitError, match="walker index"):
            engine.get_tau(walker=4)
        trm.run_transit(profs[2], n)           # a single-profile call: the plain getter again
        tau, last = engine.get_tau()
        np.testing.assert_allclose(tau, o.run(profs[2], want_tau=True)[1], rtol=RTOL, atol=1e-300)
    finally:
        trm.free_memory()


@pytest.mark.parametrize("nw", [1, 3, 17])
def test_batch_matches_oracle(small_case, nw):
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    engine.init(c.tcfg)
    try:
        profs = walkers(c, nw)
        spec, ok = engine.run_batch(profs, want_ok=True)
        assert ok.all()
        ref = orc.OracleEngine(c.tcfg).run_batch(profs)
        np.testing.assert_allclose(spec, ref, rtol=RTOL)
    finally:
        trm.free_memory()


def test_demo_shape_one_molecule(demo_case):
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = demo_case
    engine.init(c.tcfg)
    try:
        assert trm.get_no_samples() == 2501
        profs = walkers(c, 2, seed=5)
        spec = engine.run_batch(profs)
        ref = orc.OracleEngine(c.tcfg).run_batch(profs)
        np.testing.assert_allclose(spec, ref, rtol=RTOL)
    finally:
        trm.free_memory()


def many_molecules(nm):
    """Keyword arguments of synth.make_case for nm table molecules."""
    mols = ("H2O", "CO", "CO2", "CH4", "NH3", "HCN", "C2H2", "TiO", "VO")[:nm]
    return dict(opmol=mols, species=("He", "H2") + mols, abund=(0.15, 0.85) + (1e-4,) * nm)


@pytest.mark.parametrize("kw", [
    dict(opmol=("H2O", "CO"), cia=False),                       # specialised kernel, no CIA
    dict(opmol=("H2O", "CO", "CH4"), raygrid=(0, 30, 60)),      # generic kernel, 3 angles
    dict(opmol=(), cia=True),                                   # CIA only, no opacity table
    dict(opmol=("H2O",), raygrid=(0, 10, 20, 30, 40, 50, 60, 70, 80)),
    dict(nlayers=37, nwave=100, toomuch=1e30),                  # ragged sizes, no early exit
    dict(nlayers=150, nwave=65),
    dict(cia=2),                                                # H2-H2 + H2-He, BART's usual pair
    dict(cia=3),                                                # three pairs: generic kernel
    dict(opmol=("H2O",), cia=2),
    dict(many=5, cia=2), dict(many=6, cia=0), dict(many=7, cia=1), dict(many=8, cia=2),
    dict(many=9, cia=2),                                        # nine table molecules: generic kernel
    dict(many=9, cia=3, nlayers=300, nwave=90),                 # layer records above 64 kB of LDS
    dict(many=8, cia=2, nlayers=300, nwave=90),                 # ... where a specialised kernel exists
    dict(many=9, cia=3, nlayers=300, nwave=90, extra_keys={"solution": "transit", "starrad": 1.145}),
    dict(many=2, cia=1, nlayers=320, nwave=70, extra_keys={"solution": "transit", "starrad": 1.145}),
])
def test_kernel_variants_match_oracle(tmp_path, kw):
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    kw = dict(kw)
    kw.setdefault("nwave", 333)
    if "many" in kw:
        kw.update(many_molecules(kw.pop("many")), tlow=400.0, thigh=3000.0, tempdelt=650.0)
    c = synth.make_case(str(tmp_path), **kw)
    engine.init(c.tcfg)
    try:
        profs = walkers(c, 3, seed=4)
        spec = engine.run_batch(profs)
        ref = orc.OracleEngine(c.tcfg).run_batch(profs)
        np.testing.assert_allclose(spec, ref, rtol=RTOL, atol=1e-300)
    finally:
        trm.free_memory()


@pytest.fixture(scope="module")
def wide_cases(tmp_path_factory):
    """Six table molecules with BART's usual two CIA pairs, eight with two."""
    from bart_amd import synth
    return {nm: synth.make_case(str(tmp_path_factory.mktemp("case_wide%d" % nm)), nlayers=61, nwave=300,
                                cia=2, tlow=400.0, thigh=3000.0, tempdelt=650.0, **many_molecules(nm))
            for nm in (6, 8)}


@pytest.mark.parametrize("integ", [0, 1, 2])
@pytest.mark.parametrize("mode", ["generic", "mono_occ", "mono_ilp", "split", "quad", "octo"])
def test_every_kernel_variant_matches_oracle(small_case, wide_cases, mode, integ):
    """The RT kernels (generic fallback, single-wave specialised, producer/consumer
    split, quad-layer with four and with eight lane rows) under each integration
    rule (0 transmittance trapezoid, 1 the Simpson hybrid of SURVEY App. A-4, 2
    trapezoid in tau; oracle/rt_oracle.c column_eclipse) on the same batches --
    four molecules + one CIA pair, six + two, eight + two -- without and with an
    opaque cloud deck (its surface term takes a different route in each kernel),
    and with `toomuch` lowered so the cut (and rule 1's padded point) falls in the
    middle of the column.  BARTRT_KERNEL is read once per process, so each variant
    runs in a child; the rule travels as BARTRT_INTEG."""
    import subprocess, sys, os
    from oracle import rt_oracle as orc
    cases = [small_case, wide_cases[6], wide_cases[8]]
    jobs = []
    for c in cases:
        profs = walkers(c, 5, seed=8)
        np.save(os.path.join(c.dir, "p.npy"), profs)
        jobs.append((os.path.join(c.dir, "p.npy"), c.tcfg, os.path.join(c.dir, "s_%s_%d.npy" % (mode, integ))))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bart_amd import engine, transit_module as trm\n"
            "for pfile, tcfg, out in %r:\n"
            "    p = np.load(pfile); engine.init(tcfg); assert trm.get_integ() == %d\n"
            "    a = engine.run_batch(p)\n"
            "    trm.set_cloudtop(-1.0); b = engine.run_batch(p)\n"
            "    np.save(out, np.array([a, b])); trm.free_memory()\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), jobs, integ))
    subprocess.check_call([sys.executable, "-c", code],
                          env=dict(os.environ, BARTRT_KERNEL=mode, BARTRT_INTEG=str(integ)), timeout=600)
    others = []
    for c, (pfile, _, out) in zip(cases, jobs):
        profs, got = np.load(pfile), np.load(out)
        o = orc.OracleEngine(c.tcfg, integ=integ)
        np.testing.assert_allclose(got[0], o.run_batch(profs), rtol=RTOL)
        o.set_cloudtop(-1.0)
        np.testing.assert_allclose(got[1], o.run_batch(profs), rtol=RTOL)
        assert not np.allclose(got[0], got[1])
        o.set_integ((integ + 1) % 3)
        others.append(np.abs(o.run_batch(profs) / got[1] - 1).max())
    assert max(others) > 1e-6          # the rules are different discretisations


@pytest.mark.parametrize("integ", [1, 2])
def test_integration_rules_outputs_and_setter(small_case, integ):
    """Rules 1 and 2 through the in-process setter (bartrt_set_integ): spectrum,
    optical depth (rule 1 integrates it by the Simpson hybrid over radius) with
    `last`, and the per-angle intensities of the generic kernel, against the oracle;
    toomuch lowered to 0.7 puts the cut (and rule 1's padded point) mid-column."""
    import re
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    cfg2 = c.tcfg + ".toomuch%d" % integ
    open(cfg2, "w").write(re.sub(r"(?m)^toomuch .*$", "toomuch 0.7", open(c.tcfg).read()))
    for cfg in (c.tcfg, cfg2):
        engine.init(cfg)
        try:
            assert trm.get_integ() == 1          # the default: App. A-4's rule
            trm.set_integ(integ)
            o = orc.OracleEngine(cfg, integ=integ)
            profs = walkers(c, 7, seed=31)
            np.testing.assert_allclose(engine.run_batch(profs), o.run_batch(profs), rtol=RTOL)
            spec = trm.run_transit(profs[0], trm.get_no_samples())
            tau, last = engine.get_tau()
            rspec, rtau, rlast = o.run(profs[0], want_tau=True)
            assert np.array_equal(last, rlast)
            np.testing.assert_allclose(tau, rtau, rtol=RTOL, atol=1e-300)
            np.testing.assert_allclose(spec, rspec, rtol=RTOL)
            inten = np.zeros((5, trm.get_no_samples()))
            trm.check(trm.lib().bartrt_get_intensity(trm._ptr(inten), 5, inten.shape[1]))
            np.testing.assert_allclose(inten, o.intensity(profs[0]), rtol=RTOL)
            with pytest.raises(trm.TransitError):
                trm.set_integ(3)
        finally:
            trm.free_memory()


@pytest.mark.parametrize("integ", [0, 1, 2])
@pytest.mark.parametrize("nlayers", [3, 4, 5, 13, 100, 209])
def test_cloud_deck_sweep_across_layer_steps(tmp_path, nlayers, integ):
    """Small batches run the quad-layer kernel (four layers per step, one per lane
    row): move the cloud deck through the column so the stop layer lands in every
    row of a step, at step edges, and above the top."""
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    c = synth.make_case(str(tmp_path), nlayers=nlayers, nwave=130)
    engine.init(c.tcfg)
    try:
        trm.set_integ(integ)
        o = orc.OracleEngine(c.tcfg, integ=integ)
        profs = walkers(c, 2, seed=6)
        lp = np.log10(c.press_bar)
        tops = np.concatenate([np.linspace(lp.min() - 0.5, lp.max() + 0.5, 12), lp[[0, -1, nlayers // 2]]])
        for ct in tops:
            trm.set_cloudtop(float(ct)); o.set_cloudtop(float(ct))
            ref = o.run_batch(profs)
            # rule 1 on a column of a dozen layers (tau steps >> 1) is a sum of Simpson panels of
            # both signs, thousands of times the result in places: samples where they cancel are
            # held to 1e-12 of the spectrum's scale instead of 1e-10 of themselves
            atol = 1e-12 * np.abs(ref).max() if integ == 1 else 1e-300
            np.testing.assert_allclose(engine.run_batch(profs), ref, rtol=RTOL, atol=atol)
    finally:
        trm.free_memory()


def test_tau_and_intensity_come_from_the_slant_kernel(small_case):
    """Under the default conventions (`integ 1`, `cut slant`) `bartrt_get_tau` / `_get_intensity` -- what `tau.dat` and
    `outintens` hold (code/cf.py:46-94) -- are written by the single-wave slant kernel on its way (its OUT build), not by
    a launch of the generic kernel; values against the oracle."""
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    engine.init(c.tcfg)
    try:
        assert trm.get_cut() == "slant" and trm.get_integ() == 1
        o = orc.OracleEngine(c.tcfg)
        prof = walkers(c, 1, seed=21)[0]
        n = trm.get_no_samples()
        spec = trm.run_transit(prof, n)
        engine.walked_begin()
        tau, last = engine.get_tau()
        assert "outputs" in engine.walked_end()[2]
        rspec, rtau, rlast = o.run(prof, want_tau=True)
        assert np.array_equal(last, rlast)
        np.testing.assert_allclose(tau, rtau, rtol=1e-9, atol=1e-300)
        inten = np.zeros((5, n))
        trm.check(trm.lib().bartrt_get_intensity(trm._ptr(inten), 5, n))
        refi = o.intensity(prof)
        np.testing.assert_allclose(inten, refi, rtol=RTOL, atol=1e-12 * np.abs(refi).max())
        np.testing.assert_allclose(spec, rspec, rtol=RTOL, atol=1e-12 * np.abs(rspec).max())
    finally:
        trm.free_memory()


def test_shards_reassemble_full_spectrum(small_case):
    """Wavenumber-block sharding: the blocks of every rank, concatenated, are the
    unsharded spectrum bit for bit (no halo)."""
    from bart_amd import engine, transit_module as trm
    c = small_case
    profs = walkers(c, 4, seed=9)
    engine.init(c.tcfg)
    full = engine.run_batch(profs)
    trm.free_memory()
    parts = []
    for r in range(3):
        engine.init(c.tcfg, shard=(r, 3), kernel_by="whole")
        lo, hi = engine.local_range()
        parts.append(engine.run_batch(profs))
        assert parts[-1].shape[1] == hi - lo
        trm.free_memory()
    assert np.array_equal(np.concatenate(parts, axis=1), full)
    # the default: every block by the kernel that fits the block (`kernel_by local`) -- the same spectrum to rounding
    parts = []
    for r in range(3):
        engine.init(c.tcfg, shard=(r, 3))
        assert trm.get_kernel_by() == "local"
        parts.append(engine.run_batch(profs))
        trm.free_memory()
    np.testing.assert_allclose(np.concatenate(parts, axis=1), full, rtol=1e-12, atol=1e-14 * np.abs(full).max())


def test_setters_cloud_scattering_radius(small_case):
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = small_case
    engine.init(c.tcfg)
    try:
        n = trm.get_no_samples()
        o = orc.OracleEngine(c.tcfg)
        prof = walkers(c, 1, seed=21)[0]
        base = trm.run_transit(prof, n)
        trm.set_cloudtop(-1.5); o.set_cloudtop(-1.5)
        cl = trm.run_transit(prof, n)
        np.testing.assert_allclose(cl, o.run(prof), rtol=RTOL)
        assert not np.allclose(cl, base)
        trm.set_scattering(1, 2.0); o.set_scattering(1, 2.0)
        np.testing.assert_allclose(trm.run_transit(prof, n), o.run(prof), rtol=RTOL)
        trm.set_scattering(2, 0.0); o.set_scattering(2, 0.0)
        np.testing.assert_allclose(trm.run_transit(prof, n), o.run(prof), rtol=RTOL)
        trm.set_radius(90000.0); o.set_radius(90000.0)
        np.testing.assert_allclose(trm.run_transit(prof, n), o.run(prof), rtol=RTOL)
    finally:
        trm.free_memory()


def test_bad_profile_flagged(small_case):
    from bart_amd import engine, transit_module as trm
    c = small_case
    engine.init(c.tcfg)
    try:
        profs = walkers(c, 3)
        profs[1, 5] = -10.0
        spec, ok = engine.run_batch(profs, want_ok=True)
        assert list(ok) == [1, 0, 1]
        assert np.all(np.isfinite(spec[[0, 2]]))
    finally:
        trm.free_memory()


def test_wavelength_keys_of_the_demo_configuration(demo_case, tmp_path):
    """The reference's demo transit configuration gives the spectral range in
    microns (`wllow 2.0`, `wlhigh 4.0`, `wlfct 1e-4`, examples/demo/
    transit_demo.cfg:17-24): same 2501-point grid, same spectrum as the
    wavenumber keys."""
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = demo_case
    keep = [l for l in open(c.tcfg).read().splitlines() if not l.startswith(("wnlow", "wnhigh"))]
    cfg2 = str(tmp_path / "transit_wl.cfg")
    with open(cfg2, "w") as f:
        f.write("\n".join(keep) + "\nwllow 2.0\nwlhigh 4.0\nwlfct 1e-4\n")
    prof = c.profiles().ravel()
    engine.init(c.tcfg)
    a = trm.run_transit(prof, trm.get_no_samples())
    wn_a = trm.get_waveno_arr(trm.get_no_samples())
    trm.free_memory()
    engine.init(cfg2)
    try:
        n = trm.get_no_samples()
        assert n == 2501
        wn_b = trm.get_waveno_arr(n)
        assert np.array_equal(wn_a, wn_b) and wn_b[0] == 2500.0 and wn_b[-1] == 5000.0
        assert np.array_equal(trm.run_transit(prof, n), a)
        o = orc.OracleEngine(cfg2)
        assert np.array_equal(o.wn, wn_b)
        np.testing.assert_allclose(a, o.run(prof), rtol=RTOL)
    finally:
        trm.free_memory()


@pytest.mark.parametrize("mode", ["", "generic", "mono_ilp", "split", "quad"])
def test_radius_ramp_cloud(tmp_path, mode):
    """cloudrad / cloudfct / cloudext of the reference's transit whitelist
    (code/makecfg.py:46-47): a grey extinction that rises linearly
    from 0 at the upper radius to cloudext at the lower one and stays there below, on
    the radii of each call's own hydrostatic solution -- every eclipse kernel and both
    transit kernels against the oracle, cloud inside the column so some layers sit in
    each of its three regimes."""
    import os, subprocess, sys
    from bart_amd import synth
    from oracle import rt_oracle as orc
    jobs = []
    for name, extra in (("ecl", {}), ("tra", {"solution": "transit", "starrad": 1.145})):
        d = str(tmp_path / name)
        c0 = synth.make_case(d, nlayers=60, nwave=200, write=False)
        r = np.sort(c0.radius_km)
        keys = dict(extra, cloudrad="%.1f,%.1f" % (r[35], r[12]), cloudfct=1e5, cloudext=3e-9)
        c = synth.make_case(d, nlayers=60, nwave=200, extra_keys=keys)
        profs = walkers(c, 4, seed=17)
        np.save(os.path.join(d, "p.npy"), profs)
        jobs.append((os.path.join(d, "p.npy"), c.tcfg, os.path.join(d, "s.npy")))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bart_amd import engine, transit_module as trm\n"
            "for pfile, tcfg, out in %r:\n"
            "    engine.init(tcfg); np.save(out, engine.run_batch(np.load(pfile))); trm.free_memory()\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), jobs))
    env = dict(os.environ)
    env.pop("BARTRT_KERNEL", None)
    if mode:
        env["BARTRT_KERNEL"] = mode
    subprocess.check_call([sys.executable, "-c", code], env=env, timeout=600)
    for pfile, tcfg, out in jobs:
        o = orc.OracleEngine(tcfg)
        assert o.c.cloud_ext == 3e-9 and o.c.cloud_rup > o.c.cloud_rdown > 0
        ref = o.run_batch(np.load(pfile))
        np.testing.assert_allclose(np.load(out), ref, rtol=RTOL)
        o.c.cloud_ext = 0.0
        assert np.abs(o.run_batch(np.load(pfile)) / ref - 1).max() > 1e-3      # the cloud matters


def test_transparent_planet(tmp_path):
    """`transparent` (code/makecfg.py:44): transit geometry without an opaque core --
    the rays below the last chord keep its transmission (unverified);
    against the oracle, and with no effect on an eclipse run."""
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    for toomuch, nlay in ((0.3, 40), (10.0, 40), (2.0, 170), (1e30, 21)):
        c = synth.make_case(str(tmp_path / ("t%g" % toomuch)), nlayers=nlay, nwave=150, toomuch=toomuch,
                            extra_keys={"solution": "transit", "starrad": 1.145, "transparent": 1})
        profs = walkers(c, 3, seed=2)
        o = orc.OracleEngine(c.tcfg)
        engine.init(c.tcfg)
        try:
            got = engine.run_batch(profs)
            # with a cloud deck: the last chord is the deck's (the matrix-tile kernel picks the deepest counted chord)
            lp = np.log10(c.press_bar)
            trm.set_cloudtop(float(lp[nlay // 3])); o.set_cloudtop(float(lp[nlay // 3]))
            np.testing.assert_allclose(engine.run_batch(profs), o.run_batch(profs), rtol=RTOL)
        finally:
            trm.free_memory()
        o = orc.OracleEngine(c.tcfg)
        assert o.c.transparent == 1
        np.testing.assert_allclose(got, o.run_batch(profs), rtol=RTOL)
        o.c.transparent = 0
        opaque = o.run_batch(profs)
        assert np.all(got <= opaque * (1 + 1e-12)) and (toomuch > 1 or np.abs(got / opaque - 1).max() > 1e-3)
    e = synth.make_case(str(tmp_path / "ecl"), nlayers=21, nwave=150, extra_keys={"transparent": 1})
    engine.init(e.tcfg)
    try:
        np.testing.assert_allclose(engine.run_batch(profs), orc.OracleEngine(e.tcfg).run_batch(profs), rtol=RTOL)
    finally:
        trm.free_memory()


@pytest.mark.parametrize("nang", [1, 2, 3, 4, 6, 7, 8, 9, 10])
def test_ray_grids_of_other_sizes(tmp_path, nang):
    """`raygrid` is free-form (examples/demo/BART_eclipse.cfg:135).  One to nine angles run the
    single-wave kernels of rules 0 and 1 instantiated for that size (csrc/rt_eclipse_angles.hip),
    ten the generic kernel; rule 2 always the generic one.  Against the oracle, with a cloud deck,
    with one walker and with a batch, and with the next batch prefetched."""
    import torch
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    grid = tuple(np.round(np.linspace(0.0, 84.0, nang), 3)) if nang > 1 else (35.0,)
    c = synth.make_case(str(tmp_path), nlayers=37, nwave=300, raygrid=grid, tlow=400.0, thigh=3000.0, tempdelt=650.0)
    engine.init(c.tcfg)
    try:
        profs = walkers(c, 12, seed=nang)
        for rule in (1, 0, 2):
            trm.set_integ(rule)
            o = orc.OracleEngine(c.tcfg, integ=rule)
            ref = o.run_batch(profs)
            tol = dict(rtol=RTOL, atol=1e-12 * np.abs(ref).max() if rule == 1 else 0.0)
            np.testing.assert_allclose(engine.run_batch(profs), ref, **tol)
            np.testing.assert_allclose(engine.run_batch(profs[:1]), ref[:1], **tol)
            d = torch.from_numpy(profs).cuda()
            a, b = d[:6].contiguous(), d[6:].contiguous()
            engine.run_batch_dev(a, next_prof=b)
            got = engine.run_batch_dev(b).cpu().numpy()
            np.testing.assert_allclose(got, ref[6:], **tol)
            trm.set_cloudtop(-1.5); o.set_cloudtop(-1.5)
            refc = o.run_batch(profs)
            np.testing.assert_allclose(engine.run_batch(profs), refc,
                                       rtol=RTOL, atol=1e-12 * np.abs(refc).max() if rule == 1 else 0.0)
            assert not np.allclose(ref, refc)
            trm.set_cloudtop(3.0); o.set_cloudtop(3.0)      # below the bottom layer: no deck
    finally:
        trm.free_memory()


@pytest.mark.parametrize("integ", [1, 0, 2])
def test_cut_on_the_slant_depth(tmp_path, integ):
    """`cut slant`: the toomuch cut on each ray's slant depth -- its own last layer per
    angle, rule 1's padded point one unit of SLANT depth further.  Through the cfg key and the
    setter, with toomuch from "the second layer" to "never", with a cloud deck, with ray grids
    with and without a vertical ray; the per-angle intensities of a single profile too.  With
    toomuch out of reach the two cuts are the same numbers; at toomuch = 10 they differ by
    about exp(-10) of the flux."""
    import re
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    for grid in ((0, 20, 40, 60, 80), (15, 45, 75)):
        c = synth.make_case(str(tmp_path / ("g%d" % len(grid))), nlayers=44, nwave=200, raygrid=grid,
                            tlow=400.0, thigh=3000.0, tempdelt=650.0, extra_keys={"cut": "slant"})
        profs = walkers(c, 7, seed=3)
        txt = open(c.tcfg).read()
        for i, tm in enumerate((1e-3, 0.3, 2.0, 10.0, 1e30)):
            cfg = c.tcfg + ".tm%d" % i
            open(cfg, "w").write(re.sub(r"(?m)^toomuch .*$", "toomuch %r" % tm, txt))
            engine.init(cfg)
            try:
                assert trm.get_cut() == "slant"
                trm.set_integ(integ)
                o = orc.OracleEngine(cfg, integ=integ)
                assert o.c.cut_slant == 1
                ref = o.run_batch(profs)
                tol = dict(rtol=RTOL, atol=1e-12 * np.abs(ref).max() if integ == 1 else 0.0)
                np.testing.assert_allclose(engine.run_batch(profs), ref, **tol)
                trm.run_transit(profs[0], trm.get_no_samples())
                inten = np.zeros((len(grid), trm.get_no_samples()))
                trm.check(trm.lib().bartrt_get_intensity(trm._ptr(inten), len(grid), inten.shape[1]))
                refi = o.intensity(profs[0])
                np.testing.assert_allclose(inten, refi, rtol=RTOL, atol=1e-12 * np.abs(refi).max() if integ == 1 else 0.0)
                lp = np.log10(c.press_bar)
                trm.set_cloudtop(float(lp[17])); o.set_cloudtop(float(lp[17]))
                refc = o.run_batch(profs)
                np.testing.assert_allclose(engine.run_batch(profs), refc, rtol=RTOL,
                                           atol=1e-12 * np.abs(refc).max() if integ == 1 else 0.0)
                # the other value of the switch
                trm.set_cut("vertical"); o.set_cut("vertical")
                assert trm.get_cut() == "vertical"
                refv = o.run_batch(profs)
                np.testing.assert_allclose(engine.run_batch(profs), refv, rtol=RTOL,
                                           atol=1e-12 * np.abs(refv).max() if integ == 1 else 0.0)
                d = np.abs(refv / refc - 1).max()
                if tm == 1e30:
                    assert d < 1e-12
                elif tm == 10.0:
                    assert 1e-7 < d < 1e-2
            finally:
                trm.free_memory()


@pytest.mark.parametrize("npairs,solution", [(1, "eclipse"), (2, "eclipse"), (1, "transit")])
def test_cia_interpolated_by_splines(tmp_path, npairs, solution):
    """`cia_interp spline`: natural cubic splines in wavenumber (at init) and in temperature
    (per layer) instead of linear interpolation.  CIA files with curvature in both directions, a coarse
    wavenumber sampling and an uneven temperature grid; one file (the second-derivative planes ride as a
    second table: the single-wave kernels run) and two (four table slots: the generic kernel); batches,
    temperatures beyond the files' range (clamped), eclipse and transit geometry.  The product solves its own
    tridiagonal systems; the oracle takes scipy's CubicSpline.  Against the linear rule the spectra move."""
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    extra = {"cia_interp": "spline"}
    if solution == "transit":
        extra.update({"solution": "transit", "starrad": 1.145})
    c = synth.make_case(str(tmp_path), nlayers=30, nwave=260, cia=npairs, extra_keys=extra)
    rng = np.random.default_rng(5)
    for n, path in enumerate(c.cia):
        s1, s2 = (("H2", "H2"), ("H2", "He"))[n]
        ct = np.array([300.0, 450.0, 700.0, 1000.0, 1300.0, 1900.0, 2400.0]) + 30.0 * n
        cw = np.arange(c.wn[0] - 30.0, c.wn[-1] + 41.0, 17.0 + 6.0 * n)
        al = (3e-6 / (1 + n)) * (ct[:, None] / 1000.0) ** 1.7 * np.exp(-((ct[:, None] - 1200.0) / 900.0) ** 2) \
            * (1.2 + np.sin(cw[None, :] / (23.0 + 5 * n))) * np.exp(0.1 * rng.normal(size=(1, len(cw))))
        synth.write_cia(path, s1, s2, ct, cw, al)
    profs = walkers(c, 6, seed=8)          # temperatures 400 .. 3000 K: both sides of the files' ranges
    profs[0, :c.temp0.size] = 250.0 + 40.0 * np.arange(c.temp0.size)
    o = orc.OracleEngine(c.tcfg)
    assert o.c.cia_spline == 1 and o.c.ncia == npairs
    ref = o.run_batch(profs)
    lin = orc.OracleEngine(c.tcfg, cia_interp="linear").run_batch(profs)
    assert np.abs(ref / lin - 1).max() > 1e-4           # the switch matters at the 1e-6 contract
    engine.init(c.tcfg)
    try:
        tol = dict(rtol=RTOL, atol=1e-12 * np.abs(ref).max())
        np.testing.assert_allclose(engine.run_batch(profs), ref, **tol)
        np.testing.assert_allclose(engine.run_batch(profs[2:3]), ref[2:3], **tol)
        for integ in ((0, 2) if solution == "eclipse" else ()):
            trm.set_integ(integ)
            np.testing.assert_allclose(engine.run_batch(profs), orc.OracleEngine(c.tcfg, integ=integ).run_batch(profs),
                                       rtol=RTOL)
    finally:
        trm.free_memory()
    # the other value, through the environment
    os.environ["BARTRT_CIA_INTERP"] = "linear"
    try:
        engine.init(c.tcfg)
        np.testing.assert_allclose(engine.run_batch(profs), lin, rtol=RTOL, atol=1e-12 * np.abs(lin).max())
    finally:
        del os.environ["BARTRT_CIA_INTERP"]
        trm.free_memory()
    bad = c.tcfg + ".bad"
    open(bad, "w").write(open(c.tcfg).read().replace("cia_interp spline", "cia_interp cubic"))
    with pytest.raises(Exception, match="cia_interp"):
        engine.init(bad)


def test_default_kernel_choice_under_cut_slant(demo_case, small_case):
    """The launcher follows the measured table under the default conventions (csrc/kernel_table.inc, written by
    tools/tune_kernels.py; rule 1, `cut slant`): by the 64-wide columns of the launch and the table-molecule count the
    all-rays layer-parallel kernel with 32 / 16 / 8 / 4 layers per step, its adjacent-rows form (rt_eclipse_qadj.hpp) or
    the single-wave kernel -- what `bartrt_kernel_choice` names is what runs; every choice against the oracle on two
    walkers' whole spectra."""
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    name_of = {"rows32": "R=32, all rays", "rows16": "R=16, all rays", "rows8": "R=8, all rays", "rows4": "R=4, all rays",
               "adj16": "qadj<R=16>", "adj8": "qadj<R=8>", "single": "rt_eclipse_simpson_slant"}
    seen = set()
    for case, counts in ((demo_case, (1, 2, 3, 4, 5, 6, 7, 9, 13)), (small_case, (1, 4, 5, 9, 13, 14, 17, 19, 20, 22, 25, 26, 30, 31, 40))):
        engine.init(case.tcfg)
        try:
            assert trm.get_cut() == "slant" and trm.get_integ() == 1
            o = orc.OracleEngine(case.tcfg)
            ncol = (trm.get_no_samples() + 63) // 64
            for n in counts:
                want = trm.lib().bartrt_kernel_choice(len(case.opmol), n * ncol).decode()
                seen.add(want)
                profs = walkers(case, n, seed=60 + n)
                engine.walked_begin()
                got = engine.run_batch(profs)
                kname = engine.walked_end()[2]
                assert name_of[want] in kname, (n, want, kname)
                ref = o.run_batch(profs[:2])
                np.testing.assert_allclose(got[:2], ref, rtol=RTOL, atol=1e-12 * np.abs(ref).max(), err_msg=kname)
        finally:
            trm.free_memory()
    assert len(seen) >= 3, seen          # the counts above do cross the table's boundaries


def test_preparation_folded_into_the_layer_parallel_kernels(demo_case, small_case, tmp_path):
    """One to four walkers under the default conventions, while the launch is one round of workgroups (<= 512): the
    layer-parallel kernels build their walker's layer records in their own prologue (csrc/prep.hpp PrepFold;
    launch_rt_folded) instead of a prep_profiles launch in front of them -- the same code, so the same bits as with
    BARTRT_FOLD=0; the walker's flag (a profile the engine cannot evaluate) and
    its hydrostatic radii still reach the caller."""
    import subprocess, sys
    from bart_amd import engine, transit_module as trm
    jobs = []
    for name, case in (("demo", demo_case), ("small", small_case)):
        profs = walkers(case, 4, seed=77)
        np.save(str(tmp_path / (name + "_p.npy")), profs)
        jobs.append((case.tcfg, str(tmp_path / (name + "_p.npy")), str(tmp_path / (name + "_plain.npy"))))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bart_amd import engine, transit_module as trm\n"
            "for tcfg, pfile, out in %r:\n"
            "    engine.init(tcfg); p = np.load(pfile)\n"
            "    np.save(out, np.concatenate([engine.run_batch(p[:n]) for n in (1, 2, 3, 4)])); trm.free_memory()\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), jobs))
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, BARTRT_FOLD="0"), timeout=600)
    for tcfg, pfile, out in jobs:
        engine.init(tcfg)
        try:
            p = np.load(pfile)
            got, folded = [], []
            for n in (1, 2, 3, 4):
                engine.walked_begin()
                got.append(engine.run_batch(p[:n]))
                kname = engine.walked_end()[2]
                folded.append("prepares its own walkers" in kname)
            assert folded[0], kname          # one walker: a single round of workgroups on either grid
            assert np.array_equal(np.concatenate(got), np.load(out))
            # the walker's flag and radii come from the folded preparation too
            bad = p[:2].copy()
            bad[1, 3] = np.nan
            spec, ok = engine.run_batch(bad, want_ok=True)
            assert ok.tolist() == [1, 0] and np.array_equal(spec[0], got[0][0])
            n = trm.get_no_samples()
            trm.run_transit(p[0], n)
            rad = np.zeros(engine.nlayers())
            trm.check(trm.lib().bartrt_get_radius(trm._ptr(rad), rad.size))
            assert np.all(np.diff(rad) > 0) and rad[0] > 1e9
        finally:
            trm.free_memory()
