"""numpy restatement of the Python half of BART's per-step callable.

TEST INFRASTRUCTURE ONLY.  Unlike the RT engine, this half of the path IS in
/root/reference and imports there, so this restatement is PINNED: the
``tests/golden/*.npz`` fixtures were produced by importing the reference
modules themselves (tests/golden/make_golden.py) and tests/test_golden.py
checks every function below against them.

Functions and the reference lines they follow:
  pt_line / xi ............ code/PT.py:589-701, 722-739
  pt_iso .................. code/PT.py:704-719
  pt_noinversion .......... code/PT.py:384-586
  pt_inversion ............ code/PT.py:157-380
  pt_adiabatic ............ code/PT.py:741-749
  pt_piette ............... code/PT.py:752-812
  step_profiles ........... code/BARTfunc.py:318-347
  bandflux ................ code/BARTfunc.py:386-396 + code/wine.py:177-199
  energy_out .............. code/BARTfunc.py:375-377
"""
from __future__ import annotations

import numpy as np
import scipy.special as sp
from scipy.ndimage import gaussian_filter1d

SIGMA_SB = 5.670374419e-8  # scipy.constants.Stefan_Boltzmann


def trapz(y, x):
    y, x = np.asarray(y, float), np.asarray(x, float)
    return np.sum(np.diff(x) * (y[1:] + y[:-1]) / 2.0)


def xi(gamma, tau):
    return (2.0 / 3) * (1 + (1. / gamma) * (1 + (0.5 * gamma * tau - 1) * np.exp(-gamma * tau))
                        + gamma * (1 - 0.5 * tau ** 2) * sp.expn(2, gamma * tau))


def pt_line(p_bar, kappa, gamma1, gamma2, alpha, beta, r_star, t_star, t_int, sma, grav,
            t_int_type="const"):
    kappa, gamma1, gamma2 = 10 ** kappa, 10 ** gamma1, 10 ** gamma2
    if t_int_type == "thorngren":
        t_eq = (r_star / (2.0 * sma)) ** 0.5 * t_star
        f = 4.0 * SIGMA_SB * t_eq ** 4
        t_int = 1.24 * t_eq * np.exp(-(np.log(f) - 0.14) ** 2 / 2.96)
    t_irr = beta * (r_star / (2.0 * sma)) ** 0.5 * t_star
    tau = kappa * (np.asarray(p_bar) * 1e6) / grav
    return (0.75 * (t_int ** 4 * (2.0 / 3.0 + tau) + t_irr ** 4 * (1 - alpha) * xi(gamma1, tau)
                    + t_irr ** 4 * alpha * xi(gamma2, tau))) ** 0.25


def pt_iso(p_bar, t):
    return np.ones(len(p_bar)) * t


def pt_noinversion(p, a1, a2, p1, p3, t3):
    p = np.asarray(p, float)
    p0 = p.min()
    t1 = t3 - (np.log(p3 / p1) / a2) ** 2.0
    t0 = t1 - (np.log(p1 / p0) / a1) ** 2.0
    if t0 < 0 or t1 < 0 or t3 < 0:
        raise ValueError("non-physical profile")
    out = np.zeros(len(p))
    m1 = (p >= p0) & (p < p1)
    m2 = (p >= p1) & (p < p3)
    m3 = (p >= p3) & (p <= p.max())
    out[m1] = (np.log(p[m1] / p0) / a1) ** 2 + t0
    out[m2] = (np.log(p[m2] / p1) / a2) ** 2 + t1
    out[m3] = t3
    return gaussian_filter1d(out, 4, mode="nearest")


def pt_inversion(p, a1, a2, p1, p2, p3, t3):
    p = np.asarray(p, float)
    p0 = p.min()
    t2 = t3 - (np.log(p3 / p2) / a2) ** 2
    t0 = t2 + (np.log(p1 / p2) / -a2) ** 2 - (np.log(p1 / p0) / a1) ** 2
    t1 = t0 + (np.log(p1 / p0) / a1) ** 2
    if t0 < 0 or t1 < 0 or t2 < 0 or t3 < 0:
        raise ValueError("non-physical profile")
    out = np.zeros(len(p))
    m1 = (p >= p0) & (p < p1)
    m2 = (p >= p1) & (p < p2)
    m3 = (p >= p2) & (p < p3)
    m4 = (p >= p3) & (p <= p.max())
    out[m1] = (np.log(p[m1] / p0) / a1) ** 2 + t0
    out[m2] = (np.log(p[m2] / p2) / -a2) ** 2 + t2
    out[m3] = (np.log(p[m3] / p2) / a2) ** 2 + t2
    out[m4] = t3
    return gaussian_filter1d(out, 4, mode="nearest")


def pt_adiabatic(p, t0, gamma, logp0):
    return t0 / (1 + (gamma - 1) / gamma * np.log(10 ** logp0 / np.asarray(p, float)))


def pt_piette(p, t0, dtbot_32, dt32_10, dt10_0, dt0_1, dt1_01, dt01_001, dt001_top):
    import scipy.interpolate as si
    p = np.asarray(p, float)
    t = np.zeros(p.shape)
    itop, ibot = np.argmin(p), np.argmax(p)
    i001, i01, i1 = (np.argmin(np.abs(p - v)) for v in (0.01, 0.1, 1))
    i0, i10, i32 = (np.argmin(np.abs(p - v)) for v in (3.2, 10, 32))
    t[i0] = t0
    t[i10] = t0 + dt10_0
    t[i32] = t[i10] + dt32_10
    t[ibot] = t[i32] + dtbot_32
    t[i1] = t0 - dt0_1
    t[i01] = t[i1] - dt1_01
    t[i001] = t[i01] - dt01_001
    t[itop] = t[i001] - dt001_top
    ii = np.array([itop, i001, i01, i1, i0, i10, i32, ibot])
    rep = si.splrep(np.log10(p[ii]), t[ii], k=1)
    t = si.splev(np.log10(p), rep)
    sig = 0.3 / np.abs(np.log10(p)[0] - np.log10(p)[1])
    return gaussian_filter1d(t, sigma=sig, mode="nearest")


def step_profiles(params, press_bar_atm, abund, species, molfit, ptargs, tmin, tmax,
                  pttype="line", t_int_type="const"):
    """-> (profiles[(S+1), L] or None, status) with status 0 ok / 1 bad
    temperature / 2 bad abundance.  press_bar_atm and abund are in atm-file
    order (bottom -> top), as BARTfunc reads them."""
    species = np.asarray(species)
    ih2 = np.where(species == "H2")[0]
    ihe = np.where(species == "He")[0]
    ratio = (abund[:, ih2] / abund[:, ihe]).squeeze()
    imetals = np.where((species != "He") & (species != "H2") & (species != "H-") &
                       (species != "e-"))[0]
    imol = [int(np.where(species == m)[0][0]) for m in molfit]
    npt = len(params) - len(molfit)
    p = np.asarray(press_bar_atm)[::-1]
    L, S = abund.shape
    prof = np.zeros((S + 1, L))
    prof[1:] = abund.T
    try:
        if pttype == "line":
            t = pt_line(p, *params[:npt], *ptargs, t_int_type)[::-1]
        else:
            fn = {"iso": pt_iso, "madhu_noinv": pt_noinversion, "madhu_inv": pt_inversion,
                  "adiabatic": pt_adiabatic, "piette": pt_piette}[pttype]
            t = fn(p, *params[:npt])[::-1]
    except ValueError:
        # the reference logs and carries on with the previous step's profile
        # (BARTfunc.py:322-324); a batch has no "previous", so it is rejected
        return prof, 1
    prof[0] = t
    prof[1:] = abund.T
    if np.any(t < tmin) or np.any(t > tmax):
        return prof, 1
    for i, m in enumerate(imol):
        prof[1 + m] = abund[:, m] * 10.0 ** params[npt + i]
    q = 1.0 - np.sum(prof[1:][imetals], axis=0)
    if np.any(q < 0.0):
        return prof, 2
    prof[1 + ih2] = ratio * q / (1.0 + ratio)
    prof[1 + ihe] = q / (1.0 + ratio)
    return prof, 0


def bandflux(spectrum, specwn, idx0, npts, nifilter, istarfl, rprs, solution="eclipse"):
    out = np.zeros(len(idx0))
    off = 0
    for f in range(len(idx0)):
        sl = slice(idx0[f], idx0[f] + npts[f])
        w = nifilter[off:off + npts[f]]
        if solution == "eclipse":
            y = (spectrum[sl] / istarfl[off:off + npts[f]]) * rprs * rprs
        else:
            y = spectrum[sl]
        out[f] = trapz(y * w, specwn[sl])
        off += npts[f]
    return out


def energy_out(spectrum, specwn, rplanet_m):
    return trapz(spectrum, specwn) * 4 * (rplanet_m * 100) ** 2
