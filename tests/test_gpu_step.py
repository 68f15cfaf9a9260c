"""GPU: the per-step converters (T(p) model, abundance scaling, rejection
sentinels, band integration) against the reference-generated golden vectors
and the pinned numpy restatement; then the whole step against the oracle chain."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def wg():
    return np.load(os.path.join(G, "wine_golden.npz"))


@pytest.fixture(scope="module")
def ptg():
    return np.load(os.path.join(G, "pt_golden.npz"))


def _setup_demo(case, wg, ptg, tmin=400.0, tmax=3000.0, molfit=("CH4",), solution=0):
    from bart_amd import engine
    engine.init(case.tcfg)
    imol = [case.species.index(m) for m in molfit]
    engine.step_setup(ptg["line_args"], tmin, tmax, case.abund0, imol, wg["demo_idx0"],
                      wg["demo_npts"], wg["demo_nifilter"], wg["demo_istarfl"],
                      float(wg["rprs"]), solution=solution)
    return imol


def test_pt_line_and_abundances_on_device(demo_case, wg, ptg):
    import torch
    import ctypes as C
    from bart_amd import engine, transit_module as trm
    from oracle import pyhalf
    c = demo_case
    _setup_demo(c, wg, ptg, tmin=0.0, tmax=1e9)
    try:
        rng = np.random.default_rng(5)
        params = np.column_stack([ptg["line_params"], rng.uniform(-2, 1, len(ptg["line_params"]))])
        d_par = torch.from_numpy(params).cuda()
        n = len(params)
        d_prof = torch.empty((n, engine.nprof()), dtype=torch.float64, device="cuda")
        d_st = torch.empty(n, dtype=torch.int32, device="cuda")
        trm.check(trm.lib().bartrt_step_profiles_dev(
            C.c_void_p(d_par.data_ptr()), n, params.shape[1], C.c_void_p(d_prof.data_ptr()),
            C.c_void_p(d_st.data_ptr()), None))
        torch.cuda.synchronize()
        prof = d_prof.cpu().numpy().reshape(n, len(c.species) + 1, -1)
        assert (d_st.cpu().numpy() == 0).all()
        for w in range(n):
            ref, st = pyhalf.step_profiles(params[w], c.press_bar, c.abund0, c.species, ["CH4"],
                                           list(ptg["line_args"]), 0.0, 1e9)
            assert st == 0
            np.testing.assert_allclose(prof[w, 0], ref[0], rtol=1e-12)   # T(p): expn, exp, pow
            np.testing.assert_allclose(prof[w, 1:], ref[1:], rtol=1e-13)
    finally:
        trm.free_memory()


@pytest.mark.parametrize("pttype,code,key", [
    ("iso", 1, "iso"), ("madhu_noinv", 2, "noinv"), ("madhu_inv", 3, "inv"),
    ("adiabatic", 4, "adiab"), ("piette", 5, "piette")])
def test_other_pt_models_on_device(demo_case, wg, ptg, pttype, code, key):
    """Parameter draws of the golden file (the reference's own T(p) for them is
    pinned on CPU in tests/test_golden.py); device vs the pinned restatement on
    this case's pressure grid, including the draws the reference rejects."""
    import torch
    import ctypes as C
    from bart_amd import engine, transit_module as trm
    from oracle import pyhalf
    c = demo_case
    engine.init(c.tcfg)
    try:
        engine.step_setup(None, 0.0, 1e9, c.abund0, [], wg["demo_idx0"], wg["demo_npts"],
                          wg["demo_nifilter"], wg["demo_istarfl"], float(wg["rprs"]), pttype=code)
        params = np.ascontiguousarray(ptg[key + "_params"])
        n = len(params)
        d_par = torch.from_numpy(params).cuda()
        d_prof = torch.empty((n, engine.nprof()), dtype=torch.float64, device="cuda")
        d_st = torch.empty(n, dtype=torch.int32, device="cuda")
        trm.check(trm.lib().bartrt_step_profiles_dev(
            C.c_void_p(d_par.data_ptr()), n, params.shape[1], C.c_void_p(d_prof.data_ptr()),
            C.c_void_p(d_st.data_ptr()), None))
        torch.cuda.synchronize()
        T = d_prof.cpu().numpy().reshape(n, len(c.species) + 1, -1)[:, 0]
        st = d_st.cpu().numpy()
        nbad = 0
        for w in range(n):
            ref, rst = pyhalf.step_profiles(params[w], c.press_bar, c.abund0, c.species, [], None,
                                            0.0, 1e9, pttype=pttype)
            assert st[w] == rst
            if rst == 0:
                np.testing.assert_allclose(T[w], ref[0], rtol=1e-12)
            nbad += rst != 0
        if key in ("noinv", "inv"):
            assert 0 < nbad < n          # both accepted and rejected draws are covered
    finally:
        trm.free_memory()


def test_bandflux_matches_reference_golden(demo_case, wg, ptg):
    """Band integration of the reference's own spectra with the reference's own
    filter weights: expected values come straight from wine.bandintegrate."""
    import torch
    import ctypes as C
    from bart_amd import transit_module as trm
    for sol, key in ((0, "demo_band_eclipse"), (2, "demo_band_direct")):
        _setup_demo(demo_case, wg, ptg, solution=sol)
        try:
            spec = torch.from_numpy(wg["demo_spectra"]).cuda()
            st = torch.zeros(3, dtype=torch.int32, device="cuda")
            band = torch.empty((3, 10), dtype=torch.float64, device="cuda")
            trm.check(trm.lib().bartrt_step_bandflux_dev(
                C.c_void_p(spec.data_ptr()), 3, C.c_void_p(st.data_ptr()),
                C.c_void_p(band.data_ptr()), None))
            torch.cuda.synchronize()
            np.testing.assert_allclose(band.cpu().numpy(), wg[key], rtol=1e-13)
        finally:
            trm.free_memory()


def test_rejection_sentinels(demo_case, wg, ptg):
    from bart_amd import engine, transit_module as trm
    _setup_demo(demo_case, wg, ptg)
    try:
        good = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5])
        hot = np.array([-1.0, -2.0, -2.0, 0.0, 1.2, -0.5])     # T > Tmax deep down
        rich = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, 4.1])      # CH4 > 1: q < 0
        band, status = engine.step_batch(np.array([good, hot, rich, good]), 10)
        assert list(status) == [0, 1, 2, 0]
        assert np.all(band[1] == -1.0) and np.all(band[2] == -1.0)     # BARTfunc.py:329,342
        assert np.all(band[0] > 0) and np.array_equal(band[0], band[3])
    finally:
        trm.free_memory()


def test_full_step_against_oracle_chain(demo_case, wg, ptg):
    from bart_amd import engine, transit_module as trm
    from oracle import pyhalf, rt_oracle as orc
    c = demo_case
    _setup_demo(c, wg, ptg)
    try:
        rng = np.random.default_rng(17)
        params = np.array([[-2.0, 0.0, 1.0, 0.0, 0.98, -0.5],
                           [-2.3, -0.4, 0.6, 0.3, 0.9, 0.4],
                           [-1.8, 0.2, 0.1, 0.7, 1.0, -1.7]])
        band, status = engine.step_batch(params, 10)
        assert (status == 0).all()
        o = orc.OracleEngine(c.tcfg)
        for w in range(len(params)):
            prof, st = pyhalf.step_profiles(params[w], c.press_bar, c.abund0, c.species, ["CH4"],
                                            list(ptg["line_args"]), 400.0, 3000.0)
            assert st == 0
            spec = o.run(prof)
            ref = pyhalf.bandflux(spec, o.wn, wg["demo_idx0"], wg["demo_npts"], wg["demo_nifilter"],
                                  wg["demo_istarfl"], float(wg["rprs"]))
            np.testing.assert_allclose(band[w], ref, rtol=1e-9)
        # eclipse depths of a hot Jupiter: 1e-5 .. 1e-2
        assert band.min() > 1e-6 and band.max() < 5e-2
    finally:
        trm.free_memory()


def test_sharded_step_equals_unsharded(demo_case, wg, ptg):
    """engine.step_batch_sharded minus the collective, on one GPU: each of three
    wavenumber-block engines in turn builds the profiles and evaluates its block;
    the concatenated blocks go through the band integration of a sharded engine
    and give the unsharded step's band fluxes."""
    import torch
    import ctypes as C
    from bart_amd import engine, transit_module as trm
    c = demo_case
    params = np.array([[-2.0, 0.0, 1.0, 0.0, 0.98, -0.5],
                       [-2.3, -0.4, 0.6, 0.3, 0.9, 0.4],
                       [-1.8, 0.2, 0.1, 0.7, 5.0, -1.7],      # rejected: T(p) above Tmax
                       [-1.8, 0.2, 0.1, 0.7, 1.0, -1.7]])
    _setup_demo(c, wg, ptg)
    try:
        band0, status0 = engine.step_batch(params, 10)
        assert list(status0) == [0, 0, 1, 0]
    finally:
        trm.free_memory()
    d_par = torch.from_numpy(params).cuda()
    n, world, blocks = len(params), 3, []
    for r in range(world):
        engine.init(c.tcfg, shard=(r, world), kernel_by="whole")
        try:
            imol = [c.species.index("CH4")]
            engine.step_setup(ptg["line_args"], 400.0, 3000.0, c.abund0, imol, wg["demo_idx0"],
                              wg["demo_npts"], wg["demo_nifilter"], wg["demo_istarfl"], float(wg["rprs"]))
            prof = torch.empty((n, engine.nprof()), dtype=torch.float64, device="cuda")
            status = torch.empty(n, dtype=torch.int32, device="cuda")
            trm.check(trm.lib().bartrt_step_profiles_dev(
                C.c_void_p(d_par.data_ptr()), n, params.shape[1], C.c_void_p(prof.data_ptr()),
                C.c_void_p(status.data_ptr()), engine._stream_ptr()))   # the stream run_batch_dev uses
            blocks.append(engine.run_batch_dev(prof).clone())
            if r == world - 1:
                spec = torch.cat(blocks, dim=1).contiguous()
                assert spec.shape == (n, trm.get_no_samples())
                band = torch.empty((n, 10), dtype=torch.float64, device="cuda")
                trm.check(trm.lib().bartrt_step_bandflux_dev(
                    C.c_void_p(spec.data_ptr()), n, C.c_void_p(status.data_ptr()),
                    C.c_void_p(band.data_ptr()), engine._stream_ptr()))
                torch.cuda.synchronize()
                assert np.array_equal(status.cpu().numpy(), status0)
                np.testing.assert_allclose(band.cpu().numpy(), band0, rtol=1e-13)
        finally:
            trm.free_memory()


def test_energy_balance_rejection(demo_case, wg, ptg):
    from bart_amd import engine, transit_module as trm
    _setup_demo(demo_case, wg, ptg)
    try:
        p = np.array([[-2.0, 0.0, 1.0, 0.0, 0.98, -0.5]])
        engine.step_set_ebalance(True, 1e300, 1.0)
        band, status = engine.step_batch(p, 10)
        assert status[0] == 0
        engine.step_set_ebalance(True, 1e-300, 1.0)                 # E_out > E_in always
        band, status = engine.step_batch(p, 10)
        assert status[0] == 3 and np.all(band == -1.0)              # BARTfunc.py:378-383
    finally:
        trm.free_memory()
