"""Golden vectors of the reference's OWN per-step loop: code/BARTfunc.py `main(comm)` is
imported from /root/reference (read-only; nothing is copied) and run, unmodified, on
synthetic inputs -- with stand-ins for what is absent from the image or the tree:

  * mpi4py              -> a stub module (MPI.DOUBLE) and an in-process communicator that
                           plays MC3's master (Bcast npars/niter, Scatter params, Gather
                           band fluxes, inf ends the run);
  * MCcubed.utils       -> the six helpers BARTfunc.py uses (parray, comm_*, msg, exit),
                           restated from MC3's published behaviour (modules/MCcubed is an
                           empty submodule);
  * transit_module      -> the eight entry points backed by oracle/rt_oracle (the RT engine
                           itself is an empty submodule; the stand-in records every
                           `profiles` array the loop hands to run_transit).

What is stored (tests/golden/worker_golden.npz) are inputs and outputs only: per case the
arguments of bart_amd.synthcfg.make_worker_case that regenerate the input files, the
parameter vectors scattered, the profile arrays the reference built (temperature model,
abundance scaling, H2/He renormalisation), the setter calls it made, and the band fluxes it
gathered (-1 rows for the steps it rejected).  The RT spectra inside are the oracle's --
this pins everything AROUND the engine to the reference's code, not the engine.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_worker_golden.py
"""
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

warnings.simplefilter("ignore")
np.float = float  # noqa  (np.zeros(2, np.int) at BARTfunc.py:130)
np.int = int      # noqa
REF = "/root/reference"
sys.dont_write_bytecode = True   # nothing is written into the (read-only) reference tree
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REF, "code"))
sys.path.insert(0, ROOT)

from bart_amd import synthcfg          # noqa: E402
from oracle import rt_oracle as orc    # noqa: E402


# ---------------------------------------------------------------- stand-ins
class Master:
    """MC3's side of the intercommunicator for one worker."""

    def __init__(self, param_sets):
        self.queue = [np.asarray(p, float) for p in param_sets]
        self.npars = len(self.queue[0])
        self.gathered, self.log = [], []

    def Get_rank(self):
        return 0

    def Barrier(self):
        pass

    def Bcast(self, array, root=0):
        buf = array[0] if isinstance(array, list) else array
        buf[:] = [self.npars, len(self.queue) + 3]
        self.log.append("bcast")

    def Scatter(self, send, recv, root=0):
        buf = recv[0] if isinstance(recv, list) else recv
        buf[:] = self.queue.pop(0) if self.queue else np.inf
        self.log.append("scatter")

    def Gather(self, send, recv, root=0):
        buf = send[0] if isinstance(send, list) else send
        self.gathered.append(np.array(buf, float))
        self.log.append("gather")

    def Disconnect(self):
        self.log.append("disconnect")


def _parray(string):
    """MC3 utils.parray: numbers if the words convert, else the words."""
    if string == "None":
        return None
    try:
        return np.asarray(string.split(), np.double)
    except ValueError:
        return string.split()


def _mc3_utils():
    m = types.ModuleType("MCcubed.utils")
    m.parray = _parray

    def comm_bcast(comm, array, mpitype=None):
        comm.Barrier()
        comm.Bcast(array if mpitype is None else [array, mpitype], root=0)

    def comm_scatter(comm, array, mpitype=None):
        comm.Barrier()
        comm.Scatter(None if mpitype is None else [None, mpitype], array, root=0)

    def comm_gather(comm, array, mpitype=None):
        comm.Barrier()
        comm.Gather(array if mpitype is None else [array, mpitype], None, root=0)

    def comm_disconnect(comm):
        if comm is not None:
            comm.Barrier()
            comm.Disconnect()

    def exit(comm=None, abort=False, message=None, comm2=None):
        raise SystemExit(message)

    m.comm_bcast, m.comm_scatter, m.comm_gather, m.comm_disconnect = comm_bcast, comm_scatter, comm_gather, comm_disconnect
    m.msg = lambda verb, text, *a, **k: None
    m.exit = exit
    return m


class TransitStandIn(types.ModuleType):
    """transit_module's eight entry points on the CPU oracle; keeps what it was given."""

    def __init__(self):
        super().__init__("transit_module")
        self.eng, self.calls, self.profiles, self.spectra = None, [], [], []

    def transit_init(self, argc, argv):
        assert argc == len(argv) == 3 and argv[:2] == ["transit", "-c"]
        self.eng = orc.OracleEngine(argv[2])

    def get_no_samples(self):
        return len(self.eng.wn)

    def get_waveno_arr(self, n):
        return np.array(self.eng.wn[:n])

    def set_radius(self, r):
        self.calls.append(("set_radius", len(self.profiles), float(r), 0.0))
        self.eng.set_radius(r)

    def set_cloudtop(self, p):
        self.calls.append(("set_cloudtop", len(self.profiles), float(p), 0.0))
        self.eng.set_cloudtop(p)

    def set_scattering(self, flag, value):
        self.calls.append(("set_scattering", len(self.profiles), float(flag), float(value)))
        self.eng.set_scattering(flag, value)

    def run_transit(self, prof, nwave):
        assert nwave == len(self.eng.wn)
        self.profiles.append(np.array(prof, float))
        s = np.array(self.eng.run(np.asarray(prof, float)))
        self.spectra.append(s)
        return s

    def free_memory(self):
        self.calls.append(("free_memory", len(self.profiles), 0.0, 0.0))


def install(trm):
    mpi4py = types.ModuleType("mpi4py")
    mpi = types.ModuleType("mpi4py.MPI")
    mpi.DOUBLE = "DOUBLE"
    mpi4py.MPI = mpi
    mc3 = types.ModuleType("MCcubed")
    mc3.utils = _mc3_utils()
    sys.modules.update({"mpi4py": mpi4py, "mpi4py.MPI": mpi, "MCcubed": mc3, "MCcubed.utils": mc3.utils,
                        "transit_module": trm})


# ---------------------------------------------------------------- the cases
def param_sets(rng, base, lo, hi, n, nPT):
    """The starting point, draws around it, and draws the loop has to reject."""
    base = np.asarray(base, float)
    out = [base]
    for _ in range(n):
        out.append(np.clip(base + rng.normal(0, 0.15, len(base)) * (hi - lo), lo, hi))
    hot = base.copy(); hot[:nPT] = [-1.0, -2.0, -2.0, 0.0, 1.2]        # T > Tmax somewhere
    cold = base.copy(); cold[:nPT] = [-5.0, 1.0, 1.0, 0.9, 0.15]       # T < Tmin somewhere
    heavy = base.copy(); heavy[-1] = 4.5                                # sum of the metals > 1
    out += [hot, out[1], cold, heavy, out[2]]
    return out


CASES = {
    "eclipse_ch4": dict(kw=dict(nwave=420, wnlow=2500.0, nlayers=30, opmol=("CH4",), molfit=("CH4",), nfilters=5,
                                params=(-2.0, 0.0, 1.0, 0.0, 0.98, -0.5)), extra={}),
    "eclipse_4mol_cloud_ray": dict(kw=dict(nwave=380, wnlow=1800.0, nlayers=28,
                                           opmol=("H2O", "CO", "CO2", "CH4"), molfit=("H2O", "CO2", "CO", "CH4"),
                                           nfilters=4, params=(-1.5, -0.8, -0.8, 0.5, 1.0, -1.0, 1.0, -0.3, 0.2, -0.5, 0.1)),
                                   extra={"cloudtop": "-1.0", "scattering": "rayleigh"}),
    "transit_2mol": dict(kw=dict(nwave=300, wnlow=3000.0, nlayers=26, opmol=("H2O", "CH4"), molfit=("CH4", "H2O"),
                                 nfilters=3, solution="transit", params=(-2.0, 0.0, 1.0, 0.0, 0.98, 97000.0, -0.4, 0.3)),
                         extra={}),
    "direct_ch4": dict(kw=dict(nwave=260, wnlow=2600.0, nlayers=24, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                               solution="direct", params=(-2.0, 0.0, 1.0, 0.0, 0.98, -0.5)), extra={}),
    # `scattering = polar`: the parameter slot is consumed, the engine gets flag 2 (BARTfunc.py:356-360)
    "eclipse_polar": dict(kw=dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                                  params=(-2.0, 0.0, 1.0, 0.0, 0.98, 0.0, -0.5)), extra={"scattering": "polar"}),
    # internal temperature from the Thorngren et al. relation instead of the `tint` constant (PT.py PT_line)
    "eclipse_thorngren": dict(kw=dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                                      params=(-2.0, 0.0, 1.0, 0.0, 0.98, -0.5)), extra={"tint_type": "thorngren"}),
    # the other temperature models through the loop (nPT = 1, 6, 3, 8 parameters in front of the molecule's)
    "pt_iso": dict(kw=dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                           params=(1500.0, -0.5)), extra={}, pttype="iso",
                   seq=[[1500.0, -0.5], [900.0, 0.3], [350.0, 0.0], [2999.0, -1.0], [3100.0, 0.0]]),
    "pt_madhu_inv": dict(kw=dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                                 params=(0.4, 0.3, 0.005, 0.1, 2.0, 1600.0, -0.5)), extra={}, pttype="madhu_inv",
                         seq=[[0.4, 0.3, 0.005, 0.1, 2.0, 1600.0, -0.5], [0.55, 0.45, 0.002, 0.3, 5.0, 1550.0, 0.2],
                              [0.3, 0.2, 0.008, 0.05, 1.0, 1650.0, -1.0], [0.4, 0.3, 0.005, 0.1, 2.0, 3300.0, 0.0]]),
    "pt_adiabatic": dict(kw=dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                                 params=(2800.0, 1.4, 2.5, -0.5)), extra={}, pttype="adiabatic",
                         seq=[[2800.0, 1.4, 2.5, -0.5], [2500.0, 1.2, 2.3, 0.2], [2900.0, 1.3, 3.0, -1.0],
                              [1500.0, 1.4, 0.0, 0.0]]),
    "pt_piette": dict(kw=dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",), nfilters=3,
                              params=(1500.0, 100.0, 100.0, 100.0, 100.0, 100.0, 100.0, 100.0, -0.5)), extra={},
                      pttype="piette",
                      seq=[[1500.0, 100.0, 100.0, 100.0, 100.0, 100.0, 100.0, 100.0, -0.5],
                           [1300.0, 250.0, 30.0, 120.0, 10.0, 80.0, 200.0, 5.0, 0.3],
                           [1900.0, 20.0, 280.0, 60.0, 150.0, 0.0, 90.0, 40.0, -1.0],
                           [1200.0, 300.0, 300.0, 300.0, 300.0, 300.0, 300.0, 300.0, 0.0]]),
    # energy balance (BARTfunc.py:365-382): a self-luminous planet (tint 800 K) on a wide orbit --
    # the TEP file's semi-major axis is chosen by the generator so that some of the steps emit
    # more than they receive
    "eclipse_ebalance": dict(kw=dict(nwave=420, wnlow=1500.0, wndelt=4.0, nlayers=26, opmol=("CH4",), molfit=("CH4",),
                                     nfilters=3, ebalance=True, params=(-3.5, 0.0, 1.0, 0.0, 0.98, -0.5)),
                             extra={"tint": "800.0"}, sma_scan=(0.3, 0.33, 0.35, 0.37, 0.38, 0.39, 0.4, 0.41, 0.42, 0.43, 0.44, 0.45, 0.46, 0.47, 0.48, 0.5, 0.52, 0.55, 0.6)),
    # PT_NoInversion raises ValueError for some draws; the loop logs it and goes on with the
    # temperature array as the previous step left it (BARTfunc.py:318-330)
    "eclipse_madhu_valueerror": dict(
        kw=dict(nwave=300, wnlow=2500.0, nlayers=26, opmol=("CH4",), molfit=("CH4",), nfilters=4,
                params=(0.5, 0.5, 1e-3, 1.0, 1500.0, -0.5)),
        extra={}, pttype="madhu_noinv",
        seq=[[0.05, 0.5, 1e-3, 1.0, 1500.0, 0.6],     # T0 < 0 on the very first step: the zeros stay
             [0.5, 0.5, 1e-3, 1.0, 1500.0, -0.5],
             [0.05, 0.5, 1e-3, 1.0, 1500.0, 0.6],     # carried profile, new abundance
             [0.6, 0.4, 3e-3, 2.0, 1700.0, 0.2],
             [0.5, 0.02, 1e-3, 1.0, 1500.0, -1.0],    # T1 < 0
             [0.6, 0.4, 3e-3, 2.0, 3400.0, 0.0],      # a valid model above Tmax: rejected ...
             [0.05, 0.5, 1e-3, 1.0, 1500.0, 0.6],     # ... and carried into this failing step
             [0.5, 0.5, 1e-3, 1.0, 1500.0, -0.5]]),
}


def main():
    trm = TransitStandIn()
    install(trm)
    import BARTfunc as ref_worker          # the reference's code/BARTfunc.py, as it is
    out = {}
    import zlib
    for name, spec in CASES.items():
        rng = np.random.default_rng([20260110, zlib.crc32(name.encode())])   # a case's draws do not depend on the others
        d = tempfile.mkdtemp(prefix="wg_")
        kw = dict(spec["kw"])
        if kw.get("solution") == "transit":
            kw["extra_keys"] = {"solution": "transit", "starrad": 1.145}   # the transit cfg makecfg would write
        case, cfg = synthcfg.make_worker_case(d, **kw)
        if spec["extra"]:
            with open(cfg, "a") as f:
                for k, v in spec["extra"].items():
                    f.write("%s = %s\n" % (k, v))
        if "pttype" in spec:
            text = open(cfg).read().replace("PTtype = line", "PTtype = " + spec["pttype"])
            open(cfg, "w").write(text)
        base = np.array(spec["kw"]["params"], float)
        nPT = 5
        if "sma_scan" in spec:
            # parameter draws that stay inside [Tmin, Tmax]; the orbit decides who fails the balance
            pars = [base] + [base + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.05, 0.3]) for _ in range(7)]
            for q in pars:
                q[3] = min(max(q[3], 0.0), 1.0)
            chosen = None
            for sma in spec["sma_scan"]:
                tepv = dict(synthcfg.HD209458B, a=sma)
                synthcfg.write_tep(os.path.join(d, "planet.tep"), tepv)
                trm.eng, trm.calls, trm.profiles, trm.spectra = None, [], [], []
                master = Master(pars)
                sys.argv = ["BARTfunc.py", "-c", cfg]
                ref_worker.main(master)
                nrej = int(sum(np.all(b == -1.0) for b in master.gathered))
                print("  sma", sma, "rejected", nrej, "of", len(pars), "run_transit calls", len(trm.profiles))
                if 0 < nrej < len(pars) and len(trm.profiles) == len(pars):
                    # every step reached the engine; the balance rejected some: keep the most even split
                    if chosen is None or abs(nrej - len(pars) / 2) < chosen[1]:
                        chosen = (sma, abs(nrej - len(pars) / 2))
            assert chosen is not None
            tepv = dict(synthcfg.HD209458B, a=chosen[0])
            synthcfg.write_tep(os.path.join(d, "planet.tep"), tepv)
            kw["tep"] = tepv
        elif "seq" in spec:
            pars = [np.array(q, float) for q in spec["seq"]]
        else:
            lo = np.where(np.abs(base) > 1e3, base * 0.98, base - 0.6)
            hi = np.where(np.abs(base) > 1e3, base * 1.02, base + 0.6)
            lo[3], hi[3] = 0.0, 1.0
            pars = param_sets(rng, base, lo, hi, 5, nPT)
        trm.eng, trm.calls, trm.profiles, trm.spectra = None, [], [], []
        master = Master(pars)
        sys.argv = ["BARTfunc.py", "-c", cfg]
        ref_worker.main(master)
        band = np.array(master.gathered)
        assert len(band) == len(pars) and master.log[-1] == "disconnect"
        accepted = np.array([i for i, b in enumerate(band) if not np.all(b == -1.0)])
        if "sma_scan" in spec:
            assert len(trm.profiles) == len(pars) > len(accepted) > 0     # rejected AFTER the engine ran
        else:
            assert len(accepted) == len(trm.profiles)
        if "seq" not in spec and "sma_scan" not in spec:
            assert len(accepted) == len(pars) - 3                       # hot, cold and heavy are rejected
        out[name + "_kw"] = np.array(json.dumps({"kw": kw, "extra": spec["extra"], "pttype": spec.get("pttype", "line")}))
        out[name + "_params"] = np.array(pars)
        out[name + "_band"] = band
        out[name + "_accepted"] = accepted
        out[name + "_profiles"] = np.array(trm.profiles)
        out[name + "_spectra"] = np.array(trm.spectra)
        out[name + "_calls"] = np.array([[c[1], c[2], c[3]] for c in trm.calls if c[0] != "free_memory"])
        out[name + "_call_names"] = np.array([c[0] for c in trm.calls])
        out[name + "_log"] = np.array(master.log)
        print(name, "steps", len(pars), "accepted", len(accepted), "band0", band[accepted[0]][:3])
    # ------------------------------------------------------------ the transit cfg as makecfg writes it
    # code/makecfg.py makeTransit(cfile, tepfile, shareOpacity): BART cfg + TEP -> the file
    # `transit -c` reads.  Its text is stored with the input directory replaced by @DIR@.
    import makecfg
    d = tempfile.mkdtemp(prefix="wg_cfg_")
    case, cfg = synthcfg.make_worker_case(d, nwave=200, wnlow=2500.0, nlayers=20, opmol=("CH4",), molfit=("CH4",),
                                          nfilters=3, cia=2)
    tkeys = dict(case.keys)
    bart = os.path.join(d, "BART_full.cfg")
    with open(bart, "w") as f:
        f.write("".join(l for l in open(cfg) if l.split("=")[0].strip() not in ("tconfig", "solution")))
        f.write("tconfig = %s\n" % os.path.join(d, "made_transit.cfg"))
        f.write("molfile = %s\n" % tkeys["molfile"])
        f.write("csfile = " + "\n         ".join(str(tkeys["csfile"]).split(",")) + "\n")
        f.write("linedb = %s\n         %s\n" % (os.path.join(d, "a.tli"), os.path.join(d, "b.tli")))
        f.write("opacityfile = %s\n" % tkeys["opacityfile"])
        for k in ("wnlow", "wnhigh", "wndelt", "wnfct", "wnosamp", "tlow", "thigh", "tempdelt", "toomuch",
                  "refpress", "nwidth"):
            f.write("%s = %s\n" % (k, tkeys[k]))
        f.write("raygrid = %s\n" % " ".join(str(x) for x in tkeys["raygrid"]))
        f.write("solution = direct\nverb = 11\nsavefiles = no\nallowq = 0.01\n")
        f.write("outspec = %s\n" % os.path.join(d, "out_spectrum.dat"))
        f.write("# keys makeTransit does not know must not reach the file:\nburnin = 10\nwalk = snooker\n")
    tepf = os.path.join(d, "planet.tep")
    makecfg.makeTransit(bart, tepf, True)
    text = open(os.path.join(d, "made_transit.cfg")).read().replace(d, "@DIR@")
    out["makecfg_transit_text"] = np.array(text)
    out["makecfg_case_kw"] = np.array(json.dumps(dict(nwave=200, wnlow=2500.0, nlayers=20, opmol=["CH4"],
                                                      molfit=["CH4"], nfilters=3, cia=2)))
    print(text)
    np.savez_compressed(os.path.join(HERE, "worker_golden.npz"), **out)
    print("written", os.path.join(HERE, "worker_golden.npz"))


if __name__ == "__main__":
    main()
