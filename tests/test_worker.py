"""The BARTfunc-compatible worker: configuration parsing (CPU) and the MC3 wire
protocol driven by an in-process fake communicator (GPU)."""
import numpy as np
import pytest


class FakeIntercomm:
    """Master side of MC3's protocol for one worker (reference
    code/BARTfunc.py:129-132,309-316,399,405): Bcast (npars, niter), then per
    step Scatter(params) / Gather(bandflux); an all-inf vector ends the run."""

    def __init__(self, param_sets):
        self.queue = [np.asarray(p, float) for p in param_sets]
        self.npars = len(self.queue[0])
        self.received = []
        self.log = []
        self.disconnected = False

    def Get_rank(self):
        return 0

    def Barrier(self):
        self.log.append("barrier")

    def Bcast(self, array, root=0):
        array[:] = [self.npars, len(self.queue) + 5]     # niter larger than the run: inf ends it
        self.log.append("bcast")

    def Scatter(self, send, recv, root=0):
        recv[:] = self.queue.pop(0) if self.queue else np.inf
        self.log.append("scatter")

    def Gather(self, send, recv, root=0):
        self.received.append(np.array(send, float))
        self.log.append("gather")

    def Disconnect(self):
        self.disconnected = True


def test_worker_config_parsing(tmp_path):
    from bart_amd import synthcfg
    from bart_amd.BARTfunc import WorkerConfig, parray
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=64, nlayers=12)
    c = WorkerConfig.from_cfg(cfg)
    assert c.tconfig == case.tcfg and c.atmfile == case.atm
    assert len(c.filters) == 10 and c.molfit == ["CH4"] and len(c.params) == 6
    assert c.PTtype == "line" and c.solution == "eclipse" and c.ebalance is False
    assert (c.Tmin, c.Tmax) == (400.0, 3000.0)
    assert parray("a b\n c") == ["a", "b", "c"] and parray("1 2.5") == [1.0, 2.5]


@pytest.mark.gpu
def test_worker_protocol_and_results(tmp_path):
    from bart_amd import BARTfunc, hostio, synthcfg
    from oracle import pyhalf, rt_oracle as orc
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=2501)
    good = [-2.0, 0.0, 1.0, 0.0, 0.98, -0.5]
    good2 = [-2.2, -0.3, 0.5, 0.4, 0.9, 0.3]
    hot = [-1.0, -2.0, -2.0, 0.0, 1.2, -0.5]
    rich = [-2.0, 0.0, 1.0, 0.0, 0.98, 4.1]
    comm = FakeIntercomm([good, hot, good2, rich])
    nbad = BARTfunc.main(comm, ["-c", cfg])
    assert comm.disconnected and nbad[1] == 1 and nbad[2] == 1
    assert [x for x in comm.log if x != "barrier"] == \
        ["bcast"] + ["scatter", "gather"] * 4 + ["scatter"]
    out = comm.received
    assert len(out) == 4 and all(o.shape == (10,) for o in out)
    assert np.all(out[1] == -1.0) and np.all(out[3] == -1.0)
    # independent chain: reference-pinned host readers -> oracle RT -> numpy band integration
    wc = BARTfunc.WorkerConfig.from_cfg(cfg)
    tep = hostio.TepFile(wc.tep_name)
    rstar = float(tep.getvalue("Rs")[0]) * hostio.Rsun
    rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
    mp = float(tep.getvalue("Mp")[0]) * hostio.Mjup
    ptargs = [rstar, float(tep.getvalue("Ts")[0]), 100.0, float(tep.getvalue("a")[0]) * hostio.AU,
              100.0 * hostio.G_NEWTON * mp / rp ** 2]
    species, press, _, abund = hostio.readatm(wc.atmfile)
    o = orc.OracleEngine(wc.tconfig)
    starfl, starwn, _, _ = hostio.readkurucz(wc.kurucz, ptargs[1], float(tep.getvalue("loggstar")[0]))
    idx0, npts, nif, ist = [], [], [], []
    for f in wc.filters:
        a, b, ind = hostio.resample(o.wn, *hostio.readfilter(f), starwn, starfl)
        idx0.append(ind[0][0]); npts.append(len(ind[0])); nif.append(a); ist.append(b)
    for par, got in ((good, out[0]), (good2, out[2])):
        prof, st = pyhalf.step_profiles(np.array(par), press, abund, species, ["CH4"], ptargs,
                                        400.0, 3000.0)
        assert st == 0
        ref = pyhalf.bandflux(o.run(prof), o.wn, idx0, npts, np.concatenate(nif),
                              np.concatenate(ist), rp / rstar)
        np.testing.assert_allclose(got, ref, rtol=1e-9)


@pytest.mark.gpu
def test_worker_batched_step_and_ebalance(tmp_path):
    from bart_amd import BARTfunc, synthcfg
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=1200, ebalance=True)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        rng = np.random.default_rng(2)
        p = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5]) + 0.05 * rng.normal(size=(33, 6))
        band = w.step(p)
        assert band.shape == (33, 10)
        one = np.array([w.step(q)[0] for q in p[:4]])
        # batch == one at a time (to rounding: the two batch sizes run different RT kernels)
        np.testing.assert_allclose(one, band[:4], rtol=1e-12)
        assert np.all(band[band[:, 0] >= 0] > 0)
    finally:
        w.close()


@pytest.mark.gpu
def test_worker_six_fitted_molecules_two_cia_pairs(tmp_path):
    """The shape of a fuller retrieval: six table molecules, all fitted, H2-H2 and
    H2-He CIA; a batch through Worker.step against the independent chain."""
    from bart_amd import BARTfunc, synthcfg, hostio
    from oracle import rt_oracle as orc, pyhalf
    from test_gpu_parity import many_molecules
    kw = many_molecules(6)
    base = (-2.0, 0.0, 1.0, 0.0, 0.98) + (0.0,) * 6
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=900, molfit=kw["opmol"], params=base,
                                          cia=2, tlow=400.0, thigh=3000.0, tempdelt=650.0, **kw)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        rng = np.random.default_rng(5)
        p = np.array(base) + np.concatenate([0.05 * rng.normal(size=(7, 5)),
                                             rng.uniform(-1.5, 1.0, size=(7, 6))], axis=1)
        band = w.step(p)
        wc = w.cfg
        tep = hostio.TepFile(wc.tep_name)
        rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
        mp = float(tep.getvalue("Mp")[0]) * hostio.Mjup
        ptargs = [w.rstar, w.tstar, 100.0, w.sma, 100.0 * hostio.G_NEWTON * mp / rp ** 2]
        species, press, _, abund = hostio.readatm(wc.atmfile)
        o = orc.OracleEngine(wc.tconfig)
        idx0, npts, nif, ist = w.windows
        for par, got in zip(p, band):
            prof, st = pyhalf.step_profiles(par, press, abund, species, list(kw["opmol"]), ptargs,
                                            400.0, 3000.0)
            assert st == 0
            ref = pyhalf.bandflux(o.run(prof), o.wn, idx0, npts, nif, ist, rp / w.rstar)
            np.testing.assert_allclose(got, ref, rtol=1e-9)
    finally:
        w.close()


class FakeGroup:
    """The workers' own communicator (MPI.COMM_WORLD of the processes MC3
    spawns), shared-memory version for worker threads."""

    def __init__(self, size):
        import threading
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size
        self.root_buf = None

    def view(self, rank):
        g = self

        class View:
            def Get_size(self):
                return g.size

            def Get_rank(self):
                return rank

            def Bcast(self, array, root=0):
                if rank == root:
                    g.root_buf = np.array(array)
                g.barrier.wait()
                array[...] = g.root_buf
                g.barrier.wait()

            def Gather(self, send, recv, root=0):
                g.slots[rank] = np.array(send, float)
                g.barrier.wait()
                if rank == root:
                    recv[...] = np.array(g.slots)
                g.barrier.wait()

            def Scatter(self, send, recv, root=0):
                if rank == root:
                    g.root_buf = np.array(send, float)
                g.barrier.wait()
                recv[...] = g.root_buf[rank]
                g.barrier.wait()

        return View()


@pytest.mark.gpu
def test_worker_group_batches_the_chains(tmp_path):
    """Four worker ranks as MC3 spawns them (one chain each): every step's four
    parameter vectors are evaluated as one batch by worker 0; each master-side
    communicator receives what a lone worker would have sent for its chain."""
    import threading
    from bart_amd import BARTfunc, synthcfg
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=600)
    rng = np.random.default_rng(8)
    base = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5])
    chains = [[base + 0.05 * rng.normal(size=6) for _ in range(3)] for _ in range(4)]
    chains[2][1] = np.array([-1.0, -2.0, -2.0, 0.0, 1.2, -0.5])      # a rejected (too hot) proposal
    # reference run: each chain through a lone worker
    lone = []
    for c in chains:
        comm = FakeIntercomm(c)
        BARTfunc.main(comm, ["-c", cfg])
        lone.append(comm.received)
    group = FakeGroup(4)
    comms = [FakeIntercomm(c) for c in chains]
    errors = []

    def run(r):
        try:
            BARTfunc.main(comms[r], ["-c", cfg], group=group.view(r))
        except Exception as e:                       # surface worker failures in the test thread
            errors.append(e)
            group.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for r in range(4):
        assert comms[r].disconnected and len(comms[r].received) == 3
        for got, want in zip(comms[r].received, lone[r]):
            assert np.array_equal(got, want)
    assert np.all(comms[2].received[1] == -1.0)


@pytest.mark.gpu
def test_valueerror_carries_the_previous_profile_on_request(tmp_path, monkeypatch):
    """BARTfunc.py:318-330: when the T(p) model raises ValueError the reference worker
    logs it and goes on with the profile array as the previous step left it -- range
    check included.  With BARTRT_CARRY_PROFILE=1 the one-chain
    worker reproduces that; the sequence of -1 / band-flux answers equals a numpy
    restatement of the reference loop, including a rejected (too hot) profile that is
    carried into the next failing step.  Without the switch such steps are rejected."""
    from bart_amd import BARTfunc, hostio, synthcfg
    from oracle import pyhalf, rt_oracle as orc
    good_a = [0.5, 0.5, 1e-3, 1.0, 1500.0, -0.5]
    good_b = [0.6, 0.4, 3e-3, 2.0, 1700.0, 0.2]
    verr_1 = [0.05, 0.5, 1e-3, 1.0, 1500.0, 0.6]       # T0 < 0: PT_NoInversion raises
    verr_2 = [0.5, 0.02, 1e-3, 1.0, 1500.0, -1.0]      # T1 < 0
    hot = [0.6, 0.4, 3e-3, 2.0, 3400.0, 0.0]           # a valid model above Tmax
    seq = [verr_1, good_a, verr_1, good_b, verr_2, hot, verr_1, good_a]
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=900, params=tuple(good_a))
    txt = open(cfg).read().replace("PTtype = line", "PTtype = madhu_noinv")
    open(cfg, "w").write(txt)
    wc = BARTfunc.WorkerConfig.from_cfg(cfg)
    # ---- numpy restatement of the reference loop (BARTfunc.py:309-399)
    tep = hostio.TepFile(wc.tep_name)
    rstar = float(tep.getvalue("Rs")[0]) * hostio.Rsun
    rp = float(tep.getvalue("Rp")[0]) * hostio.Rjup
    species, press, _, abund = hostio.readatm(wc.atmfile)
    o = orc.OracleEngine(wc.tconfig)
    starfl, starwn, _, _ = hostio.readkurucz(wc.kurucz, float(tep.getvalue("Ts")[0]),
                                             float(tep.getvalue("loggstar")[0]))
    idx0, npts, nif, ist = [], [], [], []
    for f in wc.filters:
        a, b, ind = hostio.resample(o.wn, *hostio.readfilter(f), starwn, starfl)
        idx0.append(ind[0][0]); npts.append(len(ind[0])); nif.append(a); ist.append(b)
    sp = list(species)
    ih2, ihe, ich4 = sp.index("H2"), sp.index("He"), sp.index("CH4")
    ratio = abund[:, ih2] / abund[:, ihe]
    imetals = [i for i, s in enumerate(sp) if s not in ("He", "H2", "H-", "e-")]
    L = len(press)
    tprofile = np.zeros(L)
    want = []
    for par in seq:
        try:
            tprofile[:] = pyhalf.pt_noinversion(np.asarray(press)[::-1], *par[:5])[::-1]
        except ValueError:
            pass                                        # "FINDME: what to do here?"
        if np.any(tprofile < wc.Tmin) or np.any(tprofile > wc.Tmax):
            want.append(-np.ones(10)); continue
        prof = np.zeros((len(sp) + 1, L))
        prof[0], prof[1:] = tprofile, abund.T
        prof[1 + ich4] = abund[:, ich4] * 10.0 ** par[5]
        q = 1.0 - prof[1:][imetals].sum(axis=0)
        if np.any(q < 0):
            want.append(-np.ones(10)); continue
        prof[1 + ih2], prof[1 + ihe] = ratio * q / (1 + ratio), q / (1 + ratio)
        want.append(pyhalf.bandflux(o.run(prof), o.wn, idx0, npts, np.concatenate(nif), np.concatenate(ist),
                                    rp / rstar))
    rejected = [bool(np.all(w == -1)) for w in want]
    assert rejected == [True, False, False, False, False, True, True, False]
    assert not np.allclose(want[1], want[2])            # a carried profile with the new abundances
    # ---- the worker, with the switch
    monkeypatch.setenv("BARTRT_CARRY_PROFILE", "1")
    comm = FakeIntercomm(seq)
    BARTfunc.main(comm, ["-c", cfg])
    assert len(comm.received) == len(seq)
    for got, ref, rej in zip(comm.received, want, rejected):
        assert bool(np.all(got == -1)) == rej
        if not rej:
            np.testing.assert_allclose(got, ref, rtol=1e-9)
    # ---- and without it: every ValueError step is a rejection
    monkeypatch.setenv("BARTRT_CARRY_PROFILE", "0")
    comm = FakeIntercomm(seq)
    BARTfunc.main(comm, ["-c", cfg])
    assert [bool(np.all(g == -1)) for g in comm.received] == [True, False, True, False, True, True, True, False]
