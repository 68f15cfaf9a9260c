"""GPU, BASELINE.json's full headline shape (100 layers x 1e4 wavenumbers, 4
molecules, 27 table temperatures, 864 MB grid): size-independent properties of
the domain, plus the oracle on slices of the grid it can finish in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_case(tmp_path_factory):
    from bart_amd import synth
    return synth.make_case(str(tmp_path_factory.mktemp("case_full")), nlayers=100, nwave=10000)


def _profiles(case, n, seed):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.make_profiles(case, n, seed)


@pytest.mark.parametrize("rule", [1, 0])
def test_full_grid_properties(full_case, rule):
    """Size-independent properties at BASELINE's full size (100 x 1e4, 864 MB grid) under the default
    integration rule (1, App. A-4) and under rule 0; the oracle follows on slices.  The closed
    forms and bounds of (1) and (3) belong to rule 0 (trapezoid in the transmittance: exact for
    isothermal columns, never above the hottest Planck function); rule 1 is held to them loosely
    -- it converges on them with the layer spacing -- and to the oracle tightly."""
    from bart_amd import engine, transit_module as trm
    from oracle import rt_oracle as orc
    c = full_case
    engine.init(c.tcfg)
    try:
        assert trm.get_integ() == 1
        trm.set_integ(rule)
        n = trm.get_no_samples()
        assert n == 10000 and engine.nlayers() == 100
        profs = _profiles(c, 12, seed=77)
        spec = engine.run_batch(profs)
        assert spec.shape == (12, 10000) and np.all(np.isfinite(spec))
        # (1) every spectrum is below the Planck flux of its hottest layer
        wn = trm.get_waveno_arr(n)
        for w in range(12):
            tmax = profs[w, :100].max()
            bmax = 2 * orc.H * wn ** 3 * orc.LS ** 2 / np.expm1(orc.H * orc.LS * wn / (orc.KB * tmax))
            if rule == 0:
                assert spec[w].min() > 0 and np.all(spec[w] <= np.pi * bmax * (1 + 1e-12))
            else:
                assert np.mean((spec[w] > 0) & (spec[w] <= np.pi * bmax * 1.02)) > 0.99
        # (2) walkers are independent: any order, any batch size, same bits
        perm = np.random.default_rng(1).permutation(12)
        assert np.array_equal(engine.run_batch(profs[perm]), spec[perm])
        # one walker takes the quad-layer kernel: same arithmetic, other schedule (rule 1: the
        # single-wave kernel adds its panels in another form, see rt_eclipse_s1.hpp)
        tol = dict(rtol=1e-13) if rule == 0 else dict(rtol=1e-10, atol=1e-12 * np.abs(spec).max())
        np.testing.assert_allclose(engine.run_batch(profs[3:4])[0], spec[3], **tol)
        big = engine.run_batch(np.tile(profs[:4], (40, 1)))          # 160 walkers: other kernel path
        np.testing.assert_allclose(big[:4], spec[:4], **tol)
        assert np.array_equal(big[4:8], big[:4])
        # (3) isothermal closed form with the engine's own optical depths
        iso = c.profiles(temp=np.full(100, 1400.0)).ravel()
        s_iso = trm.run_transit(iso, n)
        tau, last = engine.get_tau()
        B = 2 * orc.H * wn ** 3 * orc.LS ** 2 / np.expm1(orc.H * orc.LS * wn / (orc.KB * 1400.0))
        ang = np.radians([0, 20, 40, 60, 80])
        edges = np.radians([0, 10, 30, 50, 70, 90])
        wgt = np.pi * np.diff(np.sin(edges) ** 2)
        assert trm.get_cut() == "slant"        # (the default) every ray ends where ITS slant depth passes toomuch
        toomuch = float(orc.read_tcfg(c.tcfg)["toomuch"])
        inside = np.arange(100)[None, :] <= last[:, None]
        closed = 0.0
        for a, wg in zip(ang, wgt):
            over = inside & (tau / np.cos(a) > toomuch)
            la = np.where(over.any(axis=1), over.argmax(axis=1), last)
            closed = closed + wg * B * (1 - np.exp(-tau[np.arange(n), la] / np.cos(a)))
        if rule == 0:
            np.testing.assert_allclose(s_iso, closed, rtol=1e-11)
        else:   # the padded zero and the Simpson panels of the last steps: a few per cent
            assert np.median(np.abs(s_iso / closed - 1)) < 0.05
        # (4) the oracle on three 150-sample slices of the full grid
        for lo in (0, 4321, 9850):
            o = orc.OracleEngine(c.tcfg, wn_lo=lo, wn_hi=lo + 150, integ=rule)
            for w in (0, 7):
                np.testing.assert_allclose(spec[w, lo:lo + 150], o.run(profs[w]), rtol=1e-10)
    finally:
        trm.free_memory()


def test_full_grid_shards_concatenate(full_case):
    """8 wavenumber blocks (the 8-GPU layout, one after the other on this GPU)
    reassemble the unsharded spectra bit for bit."""
    from bart_amd import engine, transit_module as trm
    c = full_case
    profs = _profiles(c, 10, seed=5)
    engine.init(c.tcfg)
    full = engine.run_batch(profs)
    trm.free_memory()
    parts = []
    for r in range(8):
        engine.init(c.tcfg, shard=(r, 8), kernel_by="whole")     # (bit for bit: every block by the unsharded run's kernel)
        assert engine.local_range() == (1250 * r, 1250 * (r + 1))
        parts.append(engine.run_batch(np.tile(profs, (8, 1)))[:10])   # 80 walkers per step, as at N = 8
        trm.free_memory()
    assert np.array_equal(np.concatenate(parts, axis=1), full)


def test_full_grid_transit_geometry(tmp_path_factory):
    """Transit geometry at the headline shape: modulation bounded by the bottom
    and top radii, walkers independent of batch order and size, blocks of the 8-GPU layout concatenate bit for bit, and the oracle
    on slices."""
    from bart_amd import engine, synth, transit_module as trm
    from oracle import rt_oracle as orc
    c = synth.make_case(str(tmp_path_factory.mktemp("case_full_transit")), nlayers=100, nwave=10000,
                        extra_keys={"solution": "transit", "starrad": 1.145})
    rstar = 1.145 * 6.95508e10
    engine.init(c.tcfg)
    try:
        profs = _profiles(c, 6, seed=21)
        spec = engine.run_batch(profs)
        assert spec.shape == (6, 10000) and np.all(np.isfinite(spec))
        import ctypes
        rad = np.zeros(100)
        for w in range(3):
            one = trm.run_transit(profs[w], 10000)
            np.testing.assert_allclose(one, spec[w], rtol=1e-13)      # one walker: the same kernel, other grid
            assert trm.lib().bartrt_get_radius(rad.ctypes.data_as(ctypes.c_void_p), 100) == 0
            assert np.all(one >= (rad.min() / rstar) ** 2 * (1 - 1e-12))
            assert np.all(one <= (rad.max() / rstar) ** 2 * (1 + 1e-12))
        perm = np.random.default_rng(2).permutation(6)
        assert np.array_equal(engine.run_batch(profs[perm]), spec[perm])
        assert np.array_equal(engine.run_batch(np.tile(profs, (20, 1)))[:6], spec)
        for lo in (0, 5000, 9872):
            o = orc.OracleEngine(c.tcfg, wn_lo=lo, wn_hi=lo + 128)
            np.testing.assert_allclose(spec[2, lo:lo + 128], o.run(profs[2]), rtol=1e-10)
    finally:
        trm.free_memory()
    parts = []
    for r in range(8):
        engine.init(c.tcfg, shard=(r, 8))
        parts.append(engine.run_batch(profs))
        trm.free_memory()
    assert np.array_equal(np.concatenate(parts, axis=1), spec)
