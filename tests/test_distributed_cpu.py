"""CPU, gloo, world_size 2: the N > 1 path of the wavenumber-block sharding.

The RT kernels need a GPU, so here each rank stands in for its kernel output
with the slice of a known full spectrum that its shard would produce; what is
exercised is everything else on the N > 1 path: the shard boundaries (the same
integer split the C++ engine uses), the uneven-block all-gather that
reassembles [nwalkers, W] on every rank, and max-over-ranks timing."""
import os
import socket
import subprocess
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
