"""CPU, gloo, world_size 2: the N > 1 path of the wavenumber-block sharding.

The RT kernels need a GPU, so here each rank stands in for its kernel output
with the slice of a known full spectrum that its shard would produce; what is
exercised is everything else on the N > 1 path: the shard boundaries (the same
integer split the C++ engine uses), the uneven-block all-gather that
reassembles [nwalkers, W] on every rank, and max-over-ranks timing."""
import os
import socket
import subprocess

import pytest
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np, torch, torch.distributed as dist
    from bart_amd import engine
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    W, n = %(W)d, 5
    full = torch.from_numpy(np.random.default_rng(1).random((n, W)))
    lo, hi = W * rank // world, W * (rank + 1) // world      # engine.hip: Engine::init
    local = full[:, lo:hi].contiguous()
    out = engine.allgather_blocks(local)
    assert out.shape == (n, W), out.shape
    assert torch.equal(out, full)
    out2 = engine.allgather_blocks(local, total=W)      # static sizes: one collective
    assert torch.equal(out2, full)
    # bench.py's N > 1 step loop: steps write into bucket slots, one asynchronous
    # gather per bucket into one of two receive buffers, finished when the buffers
    # come round again; 7 steps leave a partly filled bucket for drain()
    for G in (1, 3, 4):
        pipe = engine.GatherPipeline(n, hi - lo, W, G, "cpu")
        done = []
        for i in range(7):
            pipe.slot(i).copy_((full * (i + 1))[:, lo:hi])
            o = pipe.submit(i)
            if o is not None:
                done += [x.clone() for x in o]
        for o in pipe.drain(6):
            done += list(o)
        assert len(done) == 7, (G, len(done))
        for i, o in enumerate(done):
            assert torch.equal(o, full * (i + 1)), (G, i)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert abs(t.item() - 0.1 * world) < 1e-12
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _run(world, W):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT, "W": W}],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_allgather_even_blocks():
    _run(2, 1000)


def test_allgather_uneven_blocks():
    _run(2, 2501)      # demo grid: 1250 + 1251 samples


def test_shard_bounds_cover_grid_without_overlap():
    for W in (2501, 2424, 10000, 7):
        for n in (1, 2, 3, 4, 8):
            if n > W:
                continue
            b = [(W * r // n, W * (r + 1) // n) for r in range(n)]
            assert b[0][0] == 0 and b[-1][1] == W
            assert all(b[i][1] == b[i + 1][0] for i in range(n - 1))
            assert all(hi > lo for lo, hi in b)


# ---------------------------------------------------------------------------
# The MC3-driven sharded worker (BARTRT_GPUS = G): BARTfunc.main with two worker
# processes, each the owner of one wavenumber block.
SHARDED_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %(root)r)
    import numpy as np, torch
    from multiprocessing.connection import Listener, Client
    from bart_amd import BARTfunc, engine

    rank, world, W, NF = int(sys.argv[1]), int(sys.argv[2]), 1001, 3
    addr = ("127.0.0.1", %(gport)d)

    class SockGroup:
        """The workers' own communicator (their MPI.COMM_WORLD): the mpi4py calls BARTfunc.main
        makes, over sockets -- rank 0 in the middle, every collective rooted there."""
        def __init__(self):
            if rank == 0:
                self.l = Listener(addr); self.c = {}
                for _ in range(world - 1):
                    c = self.l.accept(); self.c[c.recv()] = c
            else:
                import time
                for _ in range(400):
                    try:
                        self.c0 = Client(addr); break
                    except OSError:
                        time.sleep(0.05)
                self.c0.send(rank)
            self.log = []
        def Get_size(self): return world
        def Get_rank(self): return rank
        def Bcast(self, a, root=0):
            assert root == 0
            self.log.append("bcast")
            if rank == 0:
                for c in self.c.values(): c.send(np.array(a))
            else: a[...] = self.c0.recv()
        def Allgather(self, send, recv):
            self.log.append("allgather")
            if rank == 0:
                recv[0] = send
                for r, c in self.c.items(): recv[r] = c.recv()
                for c in self.c.values(): c.send(np.array(recv))
            else:
                self.c0.send(np.array(send)); recv[...] = self.c0.recv()
        def Gather(self, send, recv, root=0):
            assert root == 0
            self.log.append("gather")
            if rank == 0:
                recv[0] = send
                for r, c in self.c.items(): recv[r] = c.recv()
            else: self.c0.send(np.array(send))
        def Scatter(self, send, recv, root=0):
            assert root == 0
            self.log.append("scatter")
            if rank == 0:
                recv[...] = send[0]
                for r, c in self.c.items(): c.send(np.array(send[r]))
            else: recv[...] = self.c0.recv()

    class FakeIntercomm:
        """MC3's side of the protocol for this worker (code/BARTfunc.py:129-132,
        309-316, 399, 405)."""
        def __init__(self, sets):
            self.queue = [np.asarray(p, float) for p in sets]; self.received = []; self.done = False
        def Get_rank(self): return rank
        def Barrier(self): pass
        def Bcast(self, a, root=0): a[:] = [len(self.queue[0]), len(self.queue) + 3]
        def Scatter(self, send, recv, root=0): recv[:] = self.queue.pop(0) if self.queue else np.inf
        def Gather(self, send, recv, root=0): self.received.append(np.array(send, float))
        def Disconnect(self): self.done = True

    def spectrum(p, lo, hi):            # the stand-in for the RT kernels: a known spectrum
        i = np.arange(lo, hi)
        return p.sum(1, keepdims=True) * 1e-3 + np.sin(0.01 * i)[None, :] * (1 + p[:, :1])

    def bands(spec):                    # and for the band integration: three window means
        return np.stack([spec[:, 0:300].mean(1), spec[:, 300:650].mean(1), spec[:, 650:W].mean(1)], 1)

    class StubWorker:
        """Owns the block [W r / n, W (r+1) / n) of the grid (Engine::setup's split): RT
        on the block, all-gather of the blocks (the product's engine.allgather_blocks
        on the shards' process group), bands on the full grid."""
        nlayers, nspecies, nfilters = 7, 3, NF
        def __init__(self, cfg, shard=None, device=None, group=None):
            self.shard, self.group, self.nbad, self.calls = shard, group, {1: 0, 2: 0, 3: 0}, 0
            assert cfg.tconfig.endswith("transit.cfg")
        def step(self, params):
            self.calls += 1
            p = np.atleast_2d(params)
            if self.shard is None:
                return bands(spectrum(p, 0, W))
            r, n = self.shard
            lo, hi = W * r // n, W * (r + 1) // n
            full = engine.allgather_blocks(torch.from_numpy(spectrum(p, lo, hi)), self.group, total=W)
            assert full.shape == (len(p), W)
            return bands(full.numpy())
        def close(self): pass

    rng = np.random.default_rng(40 + rank)
    mine = [rng.normal(size=4) for _ in range(5)]               # this worker's chain
    lone = FakeIntercomm(mine)
    BARTfunc.main(lone, ["-c", %(cfg)r], worker_factory=StubWorker)   # reference: a lone worker
    os.environ["BARTRT_GPUS"] = str(world)
    os.environ["BARTRT_PORT"] = "%(tport)d"
    comm, group = FakeIntercomm(mine), SockGroup()
    made = []
    def factory(cfg, shard=None, device=None, group=None):
        made.append(StubWorker(cfg, shard, device, group)); return made[-1]
    BARTfunc.main(comm, ["-c", %(cfg)r], group=group, worker_factory=factory, shard_backend="gloo")
    assert comm.done and len(comm.received) == 5 and made[0].shard == (rank, world) and made[0].calls == 5
    assert made[0].group is not None
    for got, want in zip(comm.received, lone.received):
        assert got.shape == (NF,) and np.allclose(got, want, rtol=1e-13, atol=0), (got, want)
    # per step: every shard owner gets the whole batch (Allgather), evaluates it, and
    # worker 0's rows are scattered back (BARTfunc.main)
    assert group.log == ["bcast", "bcast"] + ["allgather", "scatter"] * 5, group.log
    assert not torch.distributed.is_initialized()              # main tore down what it brought up
    print("rank", rank, "ok")
''')


@pytest.mark.parametrize("world", [2, 8])
def test_mc3_driven_sharded_worker(tmp_path, world):
    """BARTfunc.main under BARTRT_GPUS=G (G = 8 is a node's worth, the 1 001-sample grid then
    splits unevenly), G worker processes as MC3 spawns them, gloo in place of RCCL and a stand-in for the GPU
    worker that returns its wavenumber block of a known spectrum: the port broadcast and
    process-group bring-up, the per-step Allgather of the chains' parameters, the
    all-gather that reassembles the spectra on every shard owner, the Scatter of the
    band fluxes -- each master-side communicator receives what a lone worker sends
    for its chain."""
    cfg = tmp_path / "BART.cfg"
    cfg.write_text("[MCMC]\ntconfig = transit.cfg\nparams = 0 0 0 0\n")
    args = {"root": ROOT, "gport": _free_port(), "tport": _free_port(), "cfg": str(cfg)}
    procs = [subprocess.Popen([sys.executable, "-c", SHARDED_WORKER % args, str(r), str(world)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env={k: v for k, v in os.environ.items() if k not in ("BARTRT_GPUS", "RANK", "WORLD_SIZE")})
             for r in range(world)]
    outs = [p.communicate(timeout=400)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o
