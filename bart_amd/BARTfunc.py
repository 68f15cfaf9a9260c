"""The per-MCMC-step callable on the MI355X engine.

Mirror of the reference worker ``code/BARTfunc.py``: same configuration keys
(its ``[MCMC]`` section, BARTfunc.py:47-123), same wire protocol with the MC3
master (BARTfunc.py:129-132, 309-316, 399, 405: broadcast of (npars, niter),
per step a scatter of ``params[npars]`` and a gather of ``bandflux[nfilters]``,
``params[0] == inf`` ends the loop), same rejection semantics (``-1`` in every
band for a non-physical T(p), a negative H2/He remainder or a violated energy
balance: BARTfunc.py:327-330, 339-344, 378-383).

What differs is where the work runs: T(p), abundance scaling, the RT engine and
the band integration are device kernels of libbartrt.so; ``Worker.step`` takes a
whole batch of walkers.  ``main(comm)`` keeps the reference's one-walker-per-
process shape so MC3 can drive it unchanged; the communicator only needs the
mpi4py methods ``Get_rank, Barrier, Bcast, Scatter, Gather, Disconnect``.
"""
from __future__ import annotations

import argparse
import configparser
import os
from dataclasses import dataclass, field

import numpy as np

from . import engine, hostio
from . import transit_module as trm

# the reference's PTfunc table (BARTfunc.py:150-155) -> device model codes
PT_NPARS = {"line": 5, "iso": 1, "madhu_noinv": 5, "madhu_inv": 6, "adiabatic": 3, "piette": 8}
PT_CODE = {"line": 0, "iso": 1, "madhu_noinv": 2, "madhu_inv": 3, "adiabatic": 4, "piette": 5}
SOLUTION_CODE = {"eclipse": 0, "transit": 1, "direct": 2}


def parray(s):
    """MC3's ``parray``: whitespace/newline separated list; numbers if they all
    parse, strings otherwise."""
    if isinstance(s, (list, tuple, np.ndarray)):
        return list(s)
    tok = str(s).split()
    try:
        return [float(t) for t in tok]
    except ValueError:
        return tok


@dataclass
class WorkerConfig:
    atmfile: str = None
    tconfig: str = None
    tep_name: str = None
    kurucz: str = None
    filters: list = field(default_factory=list)
    params: list = field(default_factory=list)
    molfit: list = field(default_factory=list)
    PTtype: str = "line"
    tint: float = 100.0
    tint_type: str = "const"
    Tmin: float = 400.0
    Tmax: float = 3000.0
    cloudtop: float = None
    scattering: str = None
    solution: str = "eclipse"
    ebalance: bool = False

    @classmethod
    def from_cfg(cls, path: str, section: str = "MCMC") -> "WorkerConfig":
        cp = configparser.ConfigParser()
        cp.optionxform = str
        if not cp.read([path]):
            raise FileNotFoundError(path)
        d = dict(cp.items(section))
        base = os.path.dirname(os.path.abspath(path))
        rel = lambda p: p if os.path.isabs(p) else os.path.normpath(os.path.join(base, p))
        c = cls()
        for k in ("atmfile", "tconfig", "tep_name", "kurucz"):
            if d.get(k):
                setattr(c, k, rel(d[k]))
        c.filters = [rel(f) for f in parray(d.get("filters", ""))]
        c.params = parray(d.get("params", ""))
        c.molfit = [str(m) for m in parray(d.get("molfit", ""))]
        c.PTtype = d.get("PTtype", c.PTtype)
        c.tint = float(d.get("tint", c.tint))
        c.tint_type = d.get("tint_type", c.tint_type)
        c.Tmin = float(d.get("Tmin", c.Tmin))
        c.Tmax = float(d.get("Tmax", c.Tmax))
        if d.get("cloudtop") not in (None, "", "None"):
            c.cloudtop = float(d["cloudtop"])
        if d.get("scattering") not in (None, "", "None"):
            c.scattering = d["scattering"]
        c.solution = d.get("solution", c.solution)
        c.ebalance = str(d.get("ebalance", "False")).strip() in ("True", "1", "true")
        return c


class Worker:
    """Initialisation of BARTfunc.py:134-299 plus a batched ``step``."""

    def __init__(self, cfg: WorkerConfig, shard=None, device=None, group=None, carry=None):
        self.cfg = cfg
        self.group = group      # process group of the wavenumber shards (None: the default group)
        if cfg.PTtype not in PT_NPARS:
            raise ValueError("unknown PTtype %r (known: %s)" % (cfg.PTtype, sorted(PT_NPARS)))
        tep = hostio.TepFile(cfg.tep_name)
        self.tstar = float(tep.getvalue("Ts")[0])
        self.rstar = float(tep.getvalue("Rs")[0]) * hostio.Rsun
        self.sma = float(tep.getvalue("a")[0]) * hostio.AU
        self.rplanet = float(tep.getvalue("Rp")[0]) * hostio.Rjup
        self.mplanet = float(tep.getvalue("Mp")[0]) * hostio.Mjup
        self.gstar = float(tep.getvalue("loggstar")[0])
        self.rprs = self.rplanet / self.rstar
        nfree = len(cfg.params)
        self.nmolfit = len(cfg.molfit)
        self.ncloud = int(cfg.cloudtop is not None)
        self.nray = int(cfg.scattering is not None)
        self.nradfit = int(cfg.solution == "transit")
        self.nPT = nfree - self.nmolfit - self.ncloud - self.nray - self.nradfit
        if self.nPT != PT_NPARS[cfg.PTtype]:
            raise ValueError("PTtype %r takes %d parameters, the configuration leaves %d"
                             % (cfg.PTtype, PT_NPARS[cfg.PTtype], self.nPT))
        self.species, press, _, self.abundances = hostio.readatm(cfg.atmfile)
        self.nlayers, self.nspecies = self.abundances.shape
        self.imol = [self.species.index(m) for m in cfg.molfit]
        gplanet = 100.0 * hostio.G_NEWTON * self.mplanet / self.rplanet ** 2
        ptargs = [self.rstar, self.tstar, cfg.tint, self.sma, gplanet]
        # engine (BARTfunc.py:226-234)
        engine.init(cfg.tconfig, shard=shard, device=device)
        self.nwave = trm.get_no_samples()
        self.specwn = trm.get_waveno_arr(self.nwave)
        self.integ = trm.get_integ()     # logged by main(): runs under different rules are not to be confused
        self.cut, self.cia_interp = trm.get_cut(), trm.get_cia_interp()
        if engine.species() != self.species or engine.nlayers() != self.nlayers:
            raise ValueError("atmfile of the MCMC configuration and 'atm' of the transit "
                             "configuration describe different atmospheres")
        # filters and star on the spectrum grid (BARTfunc.py:244-292)
        self.nfilters = len(cfg.filters)
        starwn = starfl = None
        if cfg.solution in ("eclipse", "transit"):
            starfl, starwn, _, _ = hostio.readkurucz(cfg.kurucz, self.tstar, self.gstar)
        idx0, npts, nif, ist = [], [], [], []
        for i, f in enumerate(cfg.filters):
            fwn, ftr = hostio.readfilter(f)
            if fwn[0] < self.specwn[0] or fwn[-1] > self.specwn[-1]:
                raise ValueError(
                    "Wavenumber array ({:.2f} - {:.2f} cm-1) does not cover the filter[{:d}] "
                    "wavenumber range ({:.2f} - {:.2f} cm-1).".format(
                        self.specwn[0], self.specwn[-1], i, fwn[0], fwn[-1]))
            if starwn is not None:
                a, b, ind = hostio.resample(self.specwn, fwn, ftr, starwn, starfl)
            else:
                a, b, ind = hostio.resample(self.specwn, fwn, ftr, fwn, ftr)
            ind = ind[0]
            idx0.append(int(ind[0])); npts.append(len(ind)); nif.append(a); ist.append(b)
        self.windows = (np.array(idx0, np.int32), np.array(npts, np.int32),
                        np.concatenate(nif) if nif else np.zeros(0),
                        np.concatenate(ist) if ist else np.zeros(0))
        engine.step_setup(ptargs, cfg.Tmin, cfg.Tmax, self.abundances, self.imol, *self.windows,
                          self.rprs, solution=SOLUTION_CODE[cfg.solution],
                          pttype=PT_CODE[cfg.PTtype],
                          tint_thorngren=(cfg.tint_type == "thorngren"))
        if self.nradfit or self.ncloud or self.nray:
            # BARTfunc.py:350-360: the reference sets these through engine-wide setters,
            # one walker per process; here they travel with each walker of the batch
            engine.step_set_extras(self.nradfit, self.ncloud, self.nray)
            if self.nray:
                trm.set_scattering(2 if "polar" in cfg.scattering else 1, 0.0)
        # BARTfunc.py:318-324: when the T(p) model raises ValueError the reference keeps the
        # previous step's profile in place.  Reproduced on request (walker w of every
        # batch is then chain w: the one-chain-per-process and the grouped main() below
        # keep that order); by default such a walker is rejected.
        if carry is None:
            carry = os.environ.get("BARTRT_CARRY_PROFILE", "0") == "1"
        self.carry = bool(carry)
        if self.carry:
            engine.step_set_carry(True)
        if cfg.ebalance:
            # BARTfunc.py:375-377
            e_in = (hostio.sig * self.tstar ** 4 * self.rstar ** 2 * np.pi * self.rplanet ** 2
                    / self.sma ** 2 * 1e7)
            engine.step_set_ebalance(True, e_in, 4 * (self.rplanet * 100) ** 2)
        self.nbad = {1: 0, 2: 0, 3: 0}

    def step(self, params: np.ndarray) -> np.ndarray:
        """params [nwalkers, npars] (or [npars]) -> bandflux [nwalkers, nfilters];
        rejected walkers carry -1 in every band."""
        p = np.ascontiguousarray(np.atleast_2d(np.asarray(params, np.double)))
        lo, hi = engine.local_range()
        if hi - lo != self.nwave:
            # wavenumber-sharded node: profiles on every rank, RT on the local
            # block, RCCL all-gather of the spectra, band integration on the full grid
            import torch
            d_par = torch.from_numpy(p).cuda()
            band_d, status_d, _ = engine.step_batch_sharded(d_par, self.nfilters, group=self.group)
            torch.cuda.synchronize()
            band, status = band_d.cpu().numpy(), status_d.cpu().numpy()
        else:
            band, status = engine.step_batch(p, self.nfilters)
        for s in status[status > 0]:
            self.nbad[int(s)] += 1
        return band

    def close(self):
        trm.free_memory()


# ---- MC3 communicator helpers (same call order as MCcubed.utils on the worker) ----
def comm_bcast(comm, array):
    comm.Barrier()
    comm.Bcast(array, root=0)


def comm_scatter(comm, array):
    comm.Barrier()
    comm.Scatter(None, array, root=0)


def comm_gather(comm, array):
    comm.Barrier()
    comm.Gather(array, None, root=0)


def comm_disconnect(comm):
    comm.Barrier()
    comm.Disconnect()


def shard_group_up(group, wrank: int, ngpu: int, backend: str = "nccl"):
    """Bring-up of the wavenumber-sharded workers (BARTRT_GPUS = G > 1): the first G
    ranks of the workers' communicator each drive one GPU.  Rank 0's rendezvous
    port (BARTRT_PORT) is broadcast over ``group`` and the G ranks join one
    torch.distributed process group (RCCL) -- or, when the process already has a
    default group (a launcher created it), a subgroup of its first G ranks.
    Returns the process group of the shards (None on the other ranks).
    ``backend`` is "gloo" in the CPU test of this path."""
    port = np.array([int(os.environ.get("BARTRT_PORT", "29533"))])
    group.Bcast(port, root=0)
    import torch.distributed as dist
    if dist.is_initialized():
        pg = dist.new_group(ranks=list(range(ngpu)), backend=backend)   # every rank makes the call
        return pg if wrank < ngpu else None
    if wrank >= ngpu:
        return None
    kw = {}
    if backend == "nccl":
        import torch
        torch.cuda.set_device(wrank)
        kw["device_id"] = torch.device("cuda", wrank)
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % int(port[0]),
                            rank=wrank, world_size=ngpu, **kw)
    return dist.group.WORLD


def main(comm, argv=None, group=None, worker_factory=None, shard_backend="nccl"):
    """The reference's ``main(comm)`` (BARTfunc.py:33-412) on the GPU engine.

    ``comm`` is the intercommunicator to the MC3 master, one worker process per
    chain as MC3 spawns them.  ``group`` is the workers' own communicator (their
    ``MPI.COMM_WORLD``): when it holds more than one rank the chains' parameter
    vectors are gathered to worker 0 every step, evaluated there as ONE batch on
    the GPU, and the band fluxes scattered back before each worker answers the
    master -- MC3 is unchanged, the engine is initialised once, and the forward
    models of a step run batched instead of one engine call per process.

    ``worker_factory(cfg, shard=, device=, group=)`` builds the object whose
    ``step(params[n, npars]) -> bandflux[n, nfilters]`` evaluates a batch (default:
    :class:`Worker`); the CPU test of the sharded path passes a stand-in."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("-c", "--config_file", required=True)
    args, _ = ap.parse_known_args(argv)
    cfg = WorkerConfig.from_cfg(args.config_file)
    make = worker_factory or Worker
    verb = comm.Get_rank() == 0
    array1 = np.zeros(2, int)
    comm_bcast(comm, array1)
    npars, niter = int(array1[0]), int(array1[1])
    nworkers = group.Get_size() if group is not None else 1
    wrank = group.Get_rank() if group is not None else 0
    if group is None:
        # one engine call per worker process, as the reference's worker makes it: the library reads the number of
        # chains (the local group of MC3's intercommunicator) to choose how `shareOpacity` is served -- fewer than five
        # processes are faster on engines of their own over one shared grid than on the chain service
        try:
            os.environ.setdefault("BARTRT_NCHAINS", str(int(comm.Get_size())))
        except Exception:      # noqa: BLE001  (a stand-in communicator without a size)
            pass
    # BARTRT_GPUS = G > 1: the first G workers each drive one GPU of the node and
    # hold one wavenumber block of the tables; every step they all evaluate the
    # whole batch on their block and reassemble the spectra with one RCCL
    # all-gather (Worker.step's sharded path).  The launch path is covered on CPU
    # by tests/test_distributed_cpu.py (gloo, two worker processes).
    ngpu = max(1, min(int(os.environ.get("BARTRT_GPUS", "1")), nworkers))
    shard_pg = None
    if ngpu > 1:
        shard_pg = shard_group_up(group, wrank, ngpu, shard_backend)
        w = make(cfg, shard=(wrank, ngpu), device=wrank, group=shard_pg) if wrank < ngpu else None
    else:
        w = make(cfg) if wrank == 0 else None   # the GPU engine lives in worker 0
    if verb and w is not None:
        print("There are {:d} layers and {:d} species.".format(w.nlayers, w.nspecies))
        if getattr(w, "integ", None) is not None:
            print("Integration rule of the eclipse geometry: integ {:d} ({})".format(
                w.integ, ("transmittance", "simpson", "trapz_tau")[w.integ]))
            if getattr(w, "cut", None) is not None:
                print("toomuch cut: {}; CIA interpolation: {}".format(w.cut, w.cia_interp))
    params = np.zeros(npars, np.double)
    nfilt = np.zeros(1, int)
    if nworkers > 1:
        if wrank == 0:
            nfilt[0] = w.nfilters
        group.Bcast(nfilt, root=0)
    allp = np.zeros((nworkers, npars), np.double) if (wrank == 0 or ngpu > 1) else None
    mine = np.zeros(int(nfilt[0]), np.double)
    while niter >= 0:
        niter -= 1
        comm_scatter(comm, params)
        if params[0] == np.inf:       # MC3 ends the run with an all-inf vector to every worker
            break
        if nworkers == 1:
            comm_gather(comm, np.ascontiguousarray(w.step(params)[0]))
            continue
        if ngpu > 1:
            group.Allgather(params, allp)      # every GPU-owning worker evaluates the whole batch
        else:
            group.Gather(params, allp, root=0)
        band = np.ascontiguousarray(w.step(allp)) if w is not None else None
        group.Scatter(band, mine, root=0)
        comm_gather(comm, mine)
    comm_disconnect(comm)
    nbad = w.nbad if w is not None else None
    if w is not None:
        w.close()
    if ngpu > 1 and shard_pg is not None:
        import torch.distributed as dist
        if shard_pg is dist.group.WORLD:
            dist.destroy_process_group()      # this function created it
    if verb and nbad is not None:
        print("Bad iterations of {} due to:".format("chain 0" if nworkers == 1 else "all chains"))
        print("  Temperature: {}".format(nbad[1]))
        print("  Abundance:   {}".format(nbad[2]))
        if cfg.ebalance:
            print("  Energy:      {}".format(nbad[3]))
    return nbad


if __name__ == "__main__":
    from mpi4py import MPI  # only needed when launched by MC3
    main(MPI.Comm.Get_parent(), group=MPI.COMM_WORLD)
