"""CPU: analytic known-answer tests that pin the RT oracle.

The reference holds no golden spectrum for the engine (its source is an empty
submodule), so these closed-form cases are what fixes the oracle's conventions.
"""
import numpy as np
import pytest

from bart_amd import synth
from oracle import rt_oracle as orc


def _engine(tmp_path, integ=None, cut=None, **kw):
    """integ: the integration rule the known answer belongs to (None: the default, rule 1);
    cut: which depth `toomuch` cuts (None: the default, each ray's slant depth)."""
    case = synth.make_case(str(tmp_path), **kw)
    return case, orc.OracleEngine(case.tcfg, integ=integ, cut=cut)


def _wgt(angles):
    a = np.asarray(angles, float)
    edges = np.concatenate([[0.0], 0.5 * (a[1:] + a[:-1]), [90.0]])
    s2 = np.sin(np.radians(edges)) ** 2
    return np.pi * np.diff(s2)


def test_zero_extinction_gives_zero_flux(tmp_path):
    """No opacity table, no CIA: tau == 0 everywhere and, with no surface term
    (convention A-4: the integrand is only the atmosphere), the flux is 0."""
    case, e = _engine(tmp_path, nwave=32, opmol=(), cia=False)
    spec, tau, last = e.run(case.profiles(), want_tau=True)
    assert np.all(tau == 0.0) and np.all(spec == 0.0)
    assert np.all(last == len(case.press_bar) - 1)


def test_table_interpolation_at_nodes_and_midpoints(tmp_path):
    """At a grid temperature the interpolated opacity is the table value bit
    for bit; midway it is the arithmetic mean."""
    case, e = _engine(tmp_path, nwave=24, nlayers=20, cia=False)
    op = orc.read_opacity(case.opacity)
    L = 20
    nd_fac = e.press / orc.KB
    for T, j in ((900.0, 5), (400.0, 0), (2900.0, 25)):
        prof = case.profiles(temp=np.full(L, T))
        ext, _ = e.extinction(prof)
        for l in (0, 7, 19):
            want = np.zeros(24)
            for m, s in enumerate(e.opmol):
                rho = prof[1 + s, l] * e.mass[s] * orc.AMU * (e.press[l] / (orc.KB * T))
                want += rho * op["kappa"][l, j, m]      # weights are exactly (1, 0)
            assert np.array_equal(ext[l], want)
    prof = case.profiles(temp=np.full(L, 950.0))
    ext, _ = e.extinction(prof)
    l = 3
    want = np.zeros(24)
    for m, s in enumerate(e.opmol):
        rho = prof[1 + s, l] * e.mass[s] * orc.AMU * (e.press[l] / (orc.KB * 950.0))
        want += rho * 0.5 * op["kappa"][l, 5, m] + rho * 0.5 * op["kappa"][l, 6, m]
    np.testing.assert_allclose(ext[l], want, rtol=1e-15)


@pytest.mark.parametrize("cut", ["vertical", "slant"])
def test_isothermal_closed_form(tmp_path, cut):
    """Isothermal column, any opacity: under rule 0 (trapezoid in the
    transmittance) I(mu) = B (1 - exp(-tau_last/mu)) exactly, so the flux tends to pi*B from
    below once tau >> 1 -- tau_last the depth of the column's cut (`cut vertical`) or of the
    ray's own (`cut slant`: the first layer whose slant depth passes toomuch).  (Rules 1 and 2
    reach it in the limit of fine layers: test_all_rules_converge_on_a_finely_layered_isothermal_column.)"""
    case, e = _engine(tmp_path, integ=0, cut=cut, nwave=40)
    T = 1500.0
    spec, tau, last = e.run(case.profiles(temp=np.full(100, T)), want_tau=True)
    wg = _wgt(e.angles)
    toomuch = float(e.keys["toomuch"])
    for i in range(0, 40, 7):
        B = orc.planck(e.wn[i], T)
        F = 0.0
        for a, w in zip(e.angles, wg):
            mu = np.cos(np.radians(a))
            la = last[i]
            if cut == "slant":
                over = np.nonzero(tau[i, :last[i] + 1] / mu > toomuch)[0]
                la = over[0] if len(over) else last[i]
            F += w * B * (1.0 - np.exp(-tau[i, la] / mu))
        assert abs(spec[i] / F - 1) < 1e-13
        assert tau[i, last[i]] > 10.0
        assert 1 - 1e-4 < spec[i] / (np.pi * B) <= 1.0


def test_intensity_never_exceeds_hottest_planck(tmp_path):
    """Rule 0: a jump of tau by >> 1 across one layer (a line core) does not push the
    intensity above the hottest layer's Planck function (rule 2 does)."""
    case = synth.make_case(str(tmp_path), nwave=8, nlayers=30, opmol=("H2O",), cia=False)
    op = orc.read_opacity(case.opacity)
    kap = np.full(op["kappa"].shape, 1e-2)
    kap[:12] = 1e7                      # layers are bottom -> top: an opaque floor at mid-column
    synth.write_opacity(case.opacity, op["ids"], op["temps"], op["press"], op["wn"], kappa=kap)
    e = orc.OracleEngine(case.tcfg, integ=0)
    prof = case.profiles()
    spec, tau, last = e.run(prof, want_tau=True)
    assert np.max(np.diff(tau[0][:last[0] + 1])) > 5.0
    bmax = np.array([orc.planck(w, prof[0].max()) for w in e.wn])
    assert np.all(spec <= np.pi * bmax * (1 + 1e-12)) and np.all(spec > 0)
    old = orc.OracleEngine(case.tcfg, integ=2).run(prof)          # trapezoid in tau
    assert np.any(old > np.pi * bmax)                              # the artefact it avoids


def test_grey_atmosphere_tau_is_column_mass(tmp_path):
    """Constant kappa, one absorber with constant mixing ratio, isothermal,
    constant mean mass: d(tau) = kappa * rho_m * dr and hydrostatic balance give
    tau(p) -> kappa * x_m * (p - p_top) / g up to the g(r) variation and the
    trapezoid error; the oracle must also agree with its own radii exactly."""
    kap = 0.37
    case = synth.make_case(str(tmp_path), nwave=8, nlayers=100, opmol=("H2O",), cia=False,
                           toomuch=1e30)
    # overwrite the table with a constant
    op = orc.read_opacity(case.opacity)
    synth.write_opacity(case.opacity, op["ids"], op["temps"], op["press"], op["wn"],
                        kappa=np.full(op["kappa"].shape, kap))
    e = orc.OracleEngine(case.tcfg, integ=0)      # tau by trapezoid over radius (rules 0 / 2)
    T = 1200.0
    prof = case.profiles(temp=np.full(100, T))
    spec, tau, last = e.run(prof, want_tau=True)
    ext, rad = e.extinction(prof)
    s = e.species.index("H2O")
    rho = prof[1 + s] * e.mass[s] * orc.AMU * e.press / (orc.KB * T)
    np.testing.assert_allclose(ext[:, 0], kap * rho, rtol=1e-14)
    # trapezoid of the oracle's own extinction over its own radii, top -> bottom
    r_top, e_top = rad[::-1], ext[::-1, 0]
    tt = np.concatenate([[0], np.cumsum(0.5 * (e_top[1:] + e_top[:-1]) * (r_top[:-1] - r_top[1:]))])
    np.testing.assert_allclose(tau[0], tt, rtol=1e-13)
    # closed form: mass-fraction * column mass; g varies by < 8 % over the column
    mu = prof[1:].T @ e.mass
    xm = prof[1 + s, 0] * e.mass[s] / mu[0]
    g0, r0 = float(e.keys["gsurf"]), float(e.keys["refradius"]) * 1e5
    closed = kap * xm * (e.press[::-1] - e.press[-1]) / g0
    assert np.all(np.abs(tau[0][5:] / closed[5:] - 1) < 0.10)       # constant-g estimate
    # with g(r) = g0 (r0/r)^2 along the oracle's radii: int dp / g, to the
    # discretisation error of 100 log-spaced layers
    ginv = (r_top / r0) ** 2 / g0
    p_top = e.press[::-1]
    closed_g = kap * xm * np.concatenate(
        [[0], np.cumsum(0.5 * (ginv[1:] + ginv[:-1]) * np.diff(p_top))])
    assert np.all(np.abs(tau[0][5:] / closed_g[5:] - 1) < 5e-3)


def test_flux_is_linear_in_planck_scaling(tmp_path):
    """Angle weights: an angle-independent intensity I gives F = pi * I (weights
    sum to pi) -- checked through the intensity output."""
    case, e = _engine(tmp_path, nwave=16)
    prof = case.profiles()
    inten = e.intensity(prof)
    spec = e.run(prof)
    np.testing.assert_allclose(spec, _wgt(e.angles) @ inten, rtol=1e-14)
    assert abs(_wgt(e.angles).sum() - np.pi) < 1e-14
    # limb darkening for a temperature rising inwards: normal ray is brightest
    assert np.all(inten[0] >= inten[-1])


def test_toomuch_cuts_the_column(tmp_path):
    case, e = _engine(tmp_path, nwave=64)
    spec, tau, last = e.run(case.profiles(), want_tau=True)
    for i in range(64):
        k = last[i]
        if k < 99:
            assert tau[i, k] > 10.0 and np.all(tau[i, :k] <= 10.0)
            assert np.all(tau[i, k:] == tau[i, k])


def test_simpson_switch_is_close_to_trapezoid(tmp_path):
    """The integration rule is a named convention; both agree to the
    discretisation error on a smooth column."""
    case = synth.make_case(str(tmp_path), nwave=16)
    a = orc.OracleEngine(case.tcfg, integ=0).run(case.profiles())
    b = orc.OracleEngine(case.tcfg, integ=1).run(case.profiles())
    assert np.max(np.abs(a / b - 1)) < 0.05 and not np.array_equal(a, b)


def _parabola_hybrid(x, y):
    """Simpson / trapezoid hybrid written independently of rt_oracle.c: with an even
    number of points the first interval is a trapezoid; every following pair of
    intervals is the exact integral of the parabola through its three points
    (numpy polyfit + polyint, not the closed-form panel weights)."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    n, res, start = len(x), 0.0, 0
    if n < 2:
        return 0.0
    if n % 2 == 0:
        res, start = 0.5 * (x[1] - x[0]) * (y[0] + y[1]), 1
    for j in range(start, n - 2, 2):
        xs = x[j:j + 3] - x[j]
        if xs[1] == 0 or xs[2] == xs[1]:
            res += 0.5 * xs[1] * (y[j] + y[j + 1]) + 0.5 * (xs[2] - xs[1]) * (y[j + 1] + y[j + 2])
            continue
        P = np.polyint(np.polyfit(xs / xs[2], y[j:j + 3], 2))
        res += (np.polyval(P, 1.0) - np.polyval(P, 0.0)) * xs[2]
    return res


@pytest.mark.parametrize("cut", ["vertical", "slant"])
@pytest.mark.parametrize("toomuch", [0.7, 10.0, 1e30])
def test_rule1_is_the_simpson_hybrid_of_appendix_a4(tmp_path, toomuch, cut):
    """Rule 1 (SURVEY.md App. A-4 as recalled) pinned against an independent numpy
    statement: tau[k] = hybrid over the layers k .. top taken from layer k upwards;
    I(mu) = (1/mu) hybrid over (tau, B exp(-tau/mu)) from the top down to `last`,
    plus one zero point one unit of tau further when a layer exists there.  `cut slant`
    (the default): `last` is the ray's own -- the first layer whose SLANT depth tau / mu
    passes toomuch -- and the padded point lies one unit of slant depth further (mu in tau)."""
    case = synth.make_case(str(tmp_path), nwave=12, nlayers=31, toomuch=toomuch)
    e = orc.OracleEngine(case.tcfg, integ=1, cut=cut)
    prof = case.profiles(temp=np.linspace(1900.0, 700.0, 31))
    spec, tau, last = e.run(prof, want_tau=True)
    inten = e.intensity(prof)
    ext, rad = e.extinction(prof)
    r, L = rad[::-1], 31
    for i in range(12):
        ecol, k1 = ext[::-1, i], last[i]
        for k in range(1, k1 + 1):
            up = np.arange(k, -1, -1)                     # from layer k up to the top
            want = _parabola_hybrid(r[up] - r[k], ecol[up])
            assert abs(tau[i, k] / want - 1) < 1e-12
        assert k1 == L - 1 or tau[i, k1] > toomuch
        for a, ang in enumerate(e.angles):
            mu = np.cos(np.radians(ang))
            ka, pad = k1, 1.0
            if cut == "slant":
                over = np.nonzero(tau[i, :k1 + 1] / mu > toomuch)[0]
                ka, pad = (over[0] if len(over) else k1), mu
            x = list(tau[i, :ka + 1])
            y = [orc.planck(e.wn[i], prof[0][L - 1 - k]) * np.exp(-tau[i, k] / mu) for k in range(ka + 1)]
            if ka + 1 < L:
                x.append(x[-1] + pad); y.append(0.0)
            assert abs(inten[a, i] / (_parabola_hybrid(x, y) / mu) - 1) < 1e-11
    # rule 2 is the plain trapezoid of the same integrand on rule 0's optical depth
    e2 = orc.OracleEngine(case.tcfg, integ=2, cut="vertical")
    _, tau2, last2 = e2.run(prof, want_tau=True)
    in2 = e2.intensity(prof)
    _, tau0, last0 = orc.OracleEngine(case.tcfg, integ=0, cut="vertical").run(prof, want_tau=True)
    assert np.array_equal(tau2, tau0) and np.array_equal(last2, last0)
    i, k1 = 5, last2[5]
    y = np.array([orc.planck(e.wn[i], prof[0][L - 1 - k]) * np.exp(-tau2[i, k]) for k in range(k1 + 1)])
    assert abs(in2[0, i] / np.sum(0.5 * (y[1:] + y[:-1]) * np.diff(tau2[i, :k1 + 1])) - 1) < 1e-12


def test_rules_agree_on_a_finely_layered_isothermal_column(tmp_path):
    """Isothermal, weak grey opacity, 200 layers: every rule converges on
    B (1 - exp(-tau/mu)); rule 0 is exact, rules 1 and 2 to their truncation error."""
    case = synth.make_case(str(tmp_path), nwave=6, nlayers=200, opmol=("H2O",), cia=False, toomuch=1e30,
                           ptop=1e-3, pbottom=10.0)
    op = orc.read_opacity(case.opacity)
    synth.write_opacity(case.opacity, op["ids"], op["temps"], op["press"], op["wn"],
                        kappa=np.full(op["kappa"].shape, 2e-2))
    prof = case.profiles(temp=np.full(200, 1300.0))
    res = {}
    for rule in (0, 1, 2):
        e = orc.OracleEngine(case.tcfg, integ=rule)
        I, (_, tau, last) = e.intensity(prof), e.run(prof, want_tau=True)
        B = orc.planck(e.wn[2], 1300.0)
        exact = np.array([B * (1 - np.exp(-tau[2, last[2]] / np.cos(np.radians(a)))) for a in e.angles])
        res[rule] = np.abs(I[:, 2] / exact - 1).max()
    assert res[0] < 1e-13 and res[1] < 2e-4 and res[2] < 2e-3 and res[1] < res[2]


def test_radpress_reference_level(tmp_path):
    case, e = _engine(tmp_path, nwave=8, opmol=(), cia=False)
    _, rad = e.extinction(case.profiles())
    i = int(np.argmin(np.abs(e.press - 0.1e6)))
    assert abs(rad[i] / (float(e.keys["refradius"]) * 1e5) - 1) < 2e-3
    assert np.all(np.diff(rad) > 0)
