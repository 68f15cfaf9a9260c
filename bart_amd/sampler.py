"""In-process batched MCMC driver for the GPU worker.

The reference runs one OS process per chain and MC3 moves 6-9 doubles per step
over MPI (code/BARTfunc.py:309-399).  With the forward model batched on the GPU
the natural shape is the opposite: ONE process per GPU evaluates every chain of
the ensemble in a single ``Worker.step`` call.  This module is that loop:
differential-evolution MCMC (ter Braak 2006; the reference's ``walk = demc``)
and its snooker variant (ter Braak & Vrugt 2008; ``walk = snooker``) over the
keys of the reference's ``[MCMC]`` section -- ``params, pmin, pmax, stepsize,
nchains, numit, burnin, data, uncert, walk, grtest`` (examples/demo/
BART_eclipse.cfg:43-102).  ``stepsize = 0`` keeps a parameter fixed; proposals
outside [pmin, pmax] and models the worker rejects (``-1`` sentinels) are
refused, as in the reference.

On a wavenumber-sharded node every rank runs this same loop with the same seed:
the proposals are identical everywhere, each rank computes its block of every
spectrum and the all-gather inside the model call reassembles them.
"""
from __future__ import annotations

import configparser
from dataclasses import dataclass

import numpy as np


@dataclass
class SamplerConfig:
    params: np.ndarray
    pmin: np.ndarray
    pmax: np.ndarray
    stepsize: np.ndarray
    data: np.ndarray
    uncert: np.ndarray
    nchains: int = 10
    numit: int = 10000
    burnin: int = 500
    walk: str = "demc"
    grtest: bool = True
    seed: int = 0

    @classmethod
    def from_cfg(cls, path: str, section: str = "MCMC") -> "SamplerConfig":
        cp = configparser.ConfigParser()
        cp.optionxform = str
        cp.read([path])
        d = dict(cp.items(section))
        arr = lambda k: np.array([float(x) for x in d[k].split()])
        return cls(params=arr("params"), pmin=arr("pmin"), pmax=arr("pmax"),
                   stepsize=arr("stepsize"), data=arr("data"), uncert=arr("uncert"),
                   nchains=int(d.get("nchains", 10)), numit=int(float(d.get("numit", 10000))),
                   burnin=int(d.get("burnin", 500)), walk=d.get("walk", "demc"),
                   grtest=d.get("grtest", "True").strip() == "True",
                   seed=int(d.get("seed", 0)))


def gelman_rubin(chains: np.ndarray) -> np.ndarray:
    """chains [nchains, nsamples, npar] -> potential scale reduction per parameter."""
    m, n, _ = chains.shape
    mean_c = chains.mean(axis=1)
    W = chains.var(axis=1, ddof=1).mean(axis=0)
    B = n * mean_c.var(axis=0, ddof=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.sqrt(((n - 1) / n * W + B / n) / W)


def _check_stepsize(stepsize):
    """MC3 reads a negative stepsize as "shared with parameter -stepsize" (its
    `stepsize` key, examples/demo/BART_eclipse.cfg:84-87); shared parameters are
    not implemented here and must not be mistaken for fixed ones."""
    if (np.asarray(stepsize) < 0).any():
        raise ValueError("stepsize < 0 (a parameter shared with another one, in MC3's convention) "
                         "is not supported: give every parameter its own stepsize, or 0 to fix it")


def run(model, cfg: SamplerConfig, log=None):
    """model(params[nchains, npars]) -> bandflux[nchains, ndata] (-1 rows = rejected).
    Returns dict(chain [nchains, nsteps, npars], chisq [nchains, nsteps], bestp,
    best_chisq, accept_rate, grstat)."""
    rng = np.random.default_rng(cfg.seed)
    _check_stepsize(cfg.stepsize)
    free = np.where(cfg.stepsize > 0)[0]
    nfree, nch = len(free), cfg.nchains
    nsteps = max(1, int(np.ceil(cfg.numit / nch)))
    npars = len(cfg.params)

    inv_unc = 1.0 / np.asarray(cfg.uncert, float)

    def chisq_of(p):
        m = np.asarray(model(p))
        bad = (m == -1.0).all(axis=1)            # the worker's rejection sentinel
        r = (m - cfg.data) * inv_unc
        c = np.einsum("ij,ij->i", r, r)
        c[bad] = np.inf
        return c

    # start: the configured point jittered by the stepsizes, inside the box
    x = np.tile(cfg.params, (nch, 1))
    x[:, free] += cfg.stepsize[free] * rng.normal(size=(nch, nfree))
    # the box applies to the free parameters; a fixed one keeps its configured
    # value even outside [pmin, pmax]
    clip_free = lambda v: np.clip(v[:, free], cfg.pmin[free], cfg.pmax[free])
    x[:, free] = clip_free(x)
    x[0] = cfg.params
    c = chisq_of(x)
    for _ in range(20):                      # re-draw chains that start on a rejected model
        bad = ~np.isfinite(c)
        if not bad.any():
            break
        xb = np.tile(cfg.params, (int(bad.sum()), 1))
        xb[:, free] += 0.1 * cfg.stepsize[free] * rng.normal(size=(len(xb), nfree))
        x[bad] = xb
        x[:, free] = clip_free(x)
        c = chisq_of(x)
    if not np.isfinite(c).any():
        raise RuntimeError("no chain starts on a physical model: check params/pmin/pmax")
    chain = np.zeros((nch, nsteps, npars))
    chis = np.zeros((nch, nsteps))
    naccept = 0
    gamma0 = 2.38 / np.sqrt(2 * max(nfree, 1))
    idx = np.arange(nch)
    snooker = cfg.walk == "snooker" and nch > 3
    pmin, pmax, step_free = cfg.pmin, cfg.pmax, cfg.stepsize[free]

    def others(excluded):
        """[B, nch] chain indices drawn uniformly from the chains not listed in
        `excluded` [B, nch, k] (distinct entries per chain), all at once."""
        k = excluded.shape[-1]
        draw = rng.integers(0, nch - k, size=excluded.shape[:-1])
        ex = np.sort(excluded, axis=-1)
        for j in range(k):                 # skip over the excluded ones in ascending order
            draw += draw >= ex[..., j]
        return draw

    # random numbers for a block of iterations at a time: the per-iteration loop
    # is then a dozen small array operations around one model call
    B = 256
    for t0 in range(0, nsteps, B):
        nb = min(B, nsteps - t0)
        me = np.broadcast_to(idx, (nb, nch))
        # a lone chain has no partner: the difference term vanishes, the jitter moves it
        r1 = others(me[..., None]) if nch > 1 else me
        r2 = others(np.stack([me, r1], axis=-1)) if nch > 2 else r1
        z = others(np.stack([me, r1, r2], axis=-1)) if snooker else None
        jit = 1e-3 * step_free * rng.normal(size=(nb, nch, nfree))
        gsn = rng.uniform(1.2, 2.2, size=(nb, nch, 1))
        logu = np.log(rng.random((nb, nch)))
        for b in range(nb):
            t = t0 + b
            xf = x[:, free]
            prop = x.copy()
            logjac = 0.0
            if snooker and t % 10 != 0:
                # snooker update: move along the line through a third chain
                xz = xf[z[b]]
                d = xf - xz
                nd = np.sqrt(np.einsum("ij,ij->i", d, d))[:, None]
                nd[nd == 0] = 1.0
                u = d / nd
                proj = np.einsum("ij,ij->i", xf[r1[b]] - xf[r2[b]], u)[:, None]
                pf = xf + gsn[b] * proj * u
                dn = pf - xz
                nd_new = np.sqrt(np.einsum("ij,ij->i", dn, dn))
                logjac = (nfree - 1) * (np.log(np.maximum(nd_new, 1e-300)) - np.log(nd[:, 0]))
            else:
                gam = 1.0 if t % 10 == 0 else gamma0
                pf = xf + gam * (xf[r1[b]] - xf[r2[b]]) + jit[b]
            prop[:, free] = pf
            inside = ((pf >= pmin[free]) & (pf <= pmax[free])).all(axis=1)
            if inside.all():
                cp = chisq_of(prop)
            else:
                cp = np.full(nch, np.inf)
                if inside.any():
                    cp[inside] = chisq_of(prop[inside])
            with np.errstate(invalid="ignore", over="ignore"):
                acc = (logu[b] < -0.5 * (cp - c) + logjac) & np.isfinite(cp)
            x[acc], c[acc] = prop[acc], cp[acc]
            naccept += int(acc.sum())
            chain[:, t], chis[:, t] = x, c
            if log is not None and (t + 1) % max(1, nsteps // 10) == 0:
                log("step %d/%d  best chisq %.4f  acceptance %.2f" % (
                    t + 1, nsteps, float(np.min(chis[:, :t + 1])), naccept / ((t + 1) * nch)))
    ib = np.unravel_index(np.argmin(chis), chis.shape)
    post = chain[:, min(cfg.burnin, nsteps - 1):]
    gr = gelman_rubin(post[:, :, free]) if cfg.grtest and post.shape[1] > 3 and nch > 1 else None
    return {"chain": chain, "chisq": chis, "bestp": chain[ib], "best_chisq": float(chis[ib]),
            "accept_rate": naccept / (nsteps * nch), "grstat": gr, "free": free}


def run_native(worker, cfg: SamplerConfig, log=None):
    """The same sampler through ``bartrt_mcmc_run`` (csrc/mcmc.hip): the loop,
    its random draws and the chi-square in C++, one batched model call per
    iteration -- about twice the iterations per second of :func:`run` at ten
    chains.  Needs an unsharded engine; same result dictionary as :func:`run`."""
    import ctypes as C
    from . import engine, transit_module as trm
    lo, hi = engine.local_range()
    if hi - lo != worker.nwave:
        raise ValueError("run_native: the engine is sharded; use run()")
    nch = cfg.nchains
    _check_stepsize(cfg.stepsize)
    nsteps = max(1, int(np.ceil(cfg.numit / nch)))
    npars = len(cfg.params)
    a = lambda v: np.ascontiguousarray(v, np.double)
    par, pmin, pmax, step = a(cfg.params), a(cfg.pmin), a(cfg.pmax), a(cfg.stepsize)
    data, unc = a(cfg.data), a(cfg.uncert)
    chain = np.zeros((nch, nsteps, npars))
    chis = np.zeros((nch, nsteps))
    nacc = C.c_long(0)
    nbad = (C.c_long * 4)()
    ptr = lambda v: v.ctypes.data_as(C.c_void_p)
    trm.check(trm.lib().bartrt_mcmc_run(
        nch, npars, C.c_long(nsteps), ptr(par), ptr(pmin), ptr(pmax), ptr(step), len(data), ptr(data),
        ptr(unc), int(cfg.walk == "snooker"), C.c_ulonglong(cfg.seed), ptr(chain), ptr(chis),
        C.cast(C.byref(nacc), C.c_void_p), C.cast(nbad, C.c_void_p)))
    for k in (1, 2, 3):
        worker.nbad[k] += int(nbad[k])
    free = np.where(cfg.stepsize > 0)[0]
    if log is not None:
        log("%d iterations of %d chains; best chisq %.4f  acceptance %.2f" % (
            nsteps, nch, float(chis.min()), nacc.value / (nsteps * nch)))
    ib = np.unravel_index(np.argmin(chis), chis.shape)
    post = chain[:, min(cfg.burnin, nsteps - 1):]
    gr = gelman_rubin(post[:, :, free]) if cfg.grtest and post.shape[1] > 3 and nch > 1 else None
    return {"chain": chain, "chisq": chis, "bestp": chain[ib], "best_chisq": float(chis[ib]),
            "accept_rate": nacc.value / (nsteps * nch), "grstat": gr, "free": free}
