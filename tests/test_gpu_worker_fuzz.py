"""Seeded random worker configurations through `Worker.step` against the
independent chain (reference-pinned T(p) / abundance restatement -> oracle RT ->
numpy band integration): geometry, table molecules and which of them are fitted,
CIA pairs, filter count, layer count, parameters drawn in the demo prior box."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MOLS = ("H2O", "CO", "CO2", "CH4", "NH3", "HCN")


@pytest.mark.parametrize("seed", range(16))
def test_random_worker_configuration(tmp_path, seed):
    from bart_amd import BARTfunc, synthcfg, hostio
    from oracle import rt_oracle as orc, pyhalf
    rng = np.random.default_rng(9000 + seed)
    nm = int(rng.integers(1, 7))
    mols = MOLS[:nm]
    nfit = int(rng.integers(0, nm + 1))
    molfit = tuple(rng.choice(mols, nfit, replace=False)) if nfit else ()
    sol = ["eclipse", "transit", "direct"][int(rng.integers(0, 3))]
    base = [-2.0, 0.0, 1.0, 0.0, 0.98] + ([97000.0] if sol == "transit" else []) + [0.0] * nfit
    kw = dict(nwave=int(rng.choice([300, 900, 2501])), opmol=mols, molfit=molfit, params=tuple(base),
              solution=sol, nfilters=int(rng.integers(1, 11)), cia=int(rng.integers(0, 3)),
              species=("He", "H2") + mols, abund=(0.15, 0.85) + (1e-4,) * nm,
              tlow=400.0, thigh=3000.0, tempdelt=650.0, nlayers=int(rng.choice([30, 100, 140])))
    if sol == "transit":
        kw["extra_keys"] = {"solution": "transit", "starrad": 1.145}
    case, cfg = synthcfg.make_worker_case(str(tmp_path), **kw)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        n = 5
        lo = np.array([-4.0, -1.5, -1.5, 0.0, 0.6]); hi = np.array([-1.0, 1.0, 1.0, 1.0, 1.15])
        p = np.tile(np.array(base), (n, 1))
        p[:, :5] = lo + (hi - lo) * rng.random((n, 5))
        if sol == "transit":
            p[:, 5] = rng.uniform(90000, 105000, n)
        if nfit:
            p[:, -nfit:] = rng.uniform(-2, 1.5, (n, nfit))
        band = w.step(p)
        tep = hostio.TepFile(w.cfg.tep_name)
        rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
        mp = float(tep.getvalue("Mp")[0]) * hostio.Mjup
        ptargs = [w.rstar, w.tstar, 100.0, w.sma, 100.0 * hostio.G_NEWTON * mp / rp ** 2]
        species, press, _, abund = hostio.readatm(w.cfg.atmfile)
        o = orc.OracleEngine(w.cfg.tconfig)
        idx0, npts, nif, ist = w.windows
        for k in range(n):
            pp = np.concatenate([p[k, :5], p[k, p.shape[1] - nfit:]]) if nfit else p[k, :5]
            prof, st = pyhalf.step_profiles(pp, press, abund, species, list(molfit), ptargs, 400.0, 3000.0)
            if st != 0:
                assert np.all(band[k] == -1.0)
                continue
            if sol == "transit":
                o.set_radius(p[k, 5])
            ref = pyhalf.bandflux(o.run(prof), o.wn, idx0, npts, nif, ist, rp / w.rstar, solution=sol)
            np.testing.assert_allclose(band[k], ref, rtol=1e-10, err_msg=str(kw))
    finally:
        w.close()
