"""Seeded random configurations against the oracle: layer / sample / walker
counts around the kernel-selection thresholds, table molecules and CIA pairs the
specialised kernels are and are not built for, ray grids with and without the
0/60-degree pair, both geometries, cloud decks and `toomuch` values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10
MOLS = ("H2O", "CO", "CO2", "CH4")


def _draw(rng, wide=False):
    kw = {}
    kw["nlayers"] = int(rng.choice([5, 13, 26, 40, 100, 117, 209, 230]))
    kw["nwave"] = int(rng.choice([2, 17, 64, 65, 200, 641, 1500, 4100]))
    if kw["nwave"] >= 1500:   # keep the synthetic table small: few layers, 4 temperatures
        kw["nlayers"] = int(rng.choice([13, 26, 40]))
        kw.update(tlow=400.0, thigh=3100.0, tempdelt=900.0)
    nm = int(rng.integers(0, 5))
    kw["opmol"] = tuple(rng.choice(MOLS, size=nm, replace=False)) if nm else ()
    kw["cia"] = bool(rng.integers(0, 2)) or nm == 0
    if wide:                      # up to nine table molecules, up to three CIA pairs
        from test_gpu_parity import many_molecules
        nm = int(rng.integers(1, 10))
        kw.update(many_molecules(nm), cia=int(rng.integers(0, 4 if nm >= 4 else 3)))   # the third pair is H2-CH4
        if kw["nwave"] < 1500:
            kw.update(tlow=400.0, thigh=3000.0, tempdelt=650.0)
    kw["raygrid"] = [(0, 20, 40, 60, 80), (0, 30, 60), (10, 35, 50, 65, 85), (0, 15, 30, 45, 60, 75),
                     (0, 20, 40, 60, 80)][int(rng.integers(0, 5))]
    kw["toomuch"] = float(rng.choice([0.5, 10.0, 20.0, 1e30]))
    geometry = "transit" if rng.random() < 0.4 else "eclipse"
    if geometry == "transit":
        kw["extra_keys"] = {"solution": "transit", "starrad": 1.145}
    nwalk = int(rng.choice([1, 2, 4, 5, 9, 10, 23]))
    cloud = float(rng.uniform(-4.0, 1.5)) if rng.random() < 0.4 else None
    return kw, geometry, nwalk, cloud


def _run(tmp_path, seed, integ=0, ramp=False, cut=None):
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    from test_gpu_parity import walkers
    rng = np.random.default_rng(1000 + seed)
    kw, geometry, nwalk, cloud = _draw(rng, wide=seed >= 48)   # seeds 0-47 keep their draws
    if ramp:   # a radius-ramp cloud somewhere in the column (and, in transit geometry, no opaque core)
        r = np.sort(synth.make_case(str(tmp_path), write=False, **kw).radius_km)
        lo, hi = sorted(rng.choice(len(r), 2, replace=False))
        kw.setdefault("extra_keys", {}).update(cloudrad="%.2f %.2f" % (r[hi], r[lo]), cloudext=10 ** rng.uniform(-10, -7))
        if geometry == "transit" and rng.random() < 0.5:
            kw["extra_keys"]["transparent"] = 1
    c = synth.make_case(str(tmp_path), **kw)
    engine.init(c.tcfg)
    try:
        trm.set_integ(integ)
        o = orc.OracleEngine(c.tcfg, integ=integ, cut=cut)
        if cut is not None:       # (None: the engine's and the oracle's default, `cut slant`)
            trm.set_cut(cut)
        if cloud is not None:
            trm.set_cloudtop(cloud); o.set_cloudtop(cloud)
        profs = walkers(c, nwalk, seed=seed)
        spec, ok = engine.run_batch(profs, want_ok=True)
        assert ok.all()
        ref = o.run_batch(profs)
        # rule 1 on coarse columns (13 layers, optical-depth steps >> 1) sums panels of alternating sign a thousand
        # times the result: there the kernel's difference form and the restatement's agree to 1e-12 of the
        # largest sample, not of each (tools/fuzz_sweep.py 200-560: five such columns, <= 8e-13 of the maximum)
        np.testing.assert_allclose(spec, ref, rtol=RTOL, atol=1e-12 * np.abs(ref).max() if integ == 1 else 1e-300,
                                   err_msg="%s %s walkers=%d cloud=%s integ=%d cut=%s" % (geometry, kw, nwalk, cloud, integ, cut))
    finally:
        trm.free_memory()


@pytest.mark.parametrize("seed", range(72))
def test_random_configuration(tmp_path, seed):
    _run(tmp_path, seed)


@pytest.mark.parametrize("seed", range(100, 136))
def test_random_configuration_rules_and_clouds(tmp_path, seed):
    """The same draws under integration rules 1 and 2 (seed mod 3 picks the rule, 0 too)
    and with a radius-ramp cloud / transparent core added to every second one; every fourth under
    `cut vertical` (the others under the default, `cut slant`)."""
    _run(tmp_path, seed, integ=seed % 3, ramp=seed % 2 == 1, cut="vertical" if seed % 4 == 0 else None)
