"""RCCL on the GPU: the N > 1 code path -- `dist.init_process_group("nccl")`
behind the stdout redirection, `all_gather_into_tensor` with async_op on RCCL's stream, the
double-buffered bucket reuse of GatherPipeline, the sharded per-step callable -- executed on
one MI355X under a world of ONE rank, so that the first multi-GPU run is not the first
execution of any of it.  Every process that talks to RCCL is a CHILD of pytest (started
before it touches the GPU itself; nothing re-execs a process that has initialised HIP)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "1")
    return e


def test_bench_collective_path_under_torchrun_world1(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --force-collective`: the
    driver's N > 1 launch shape with one rank.  One JSON line, its spectra held to the oracle by the line's
    own `parity` record, the N > 1 diagnostics present, the replicas comparison included."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--force-collective", "--force-replicas-leg", "--steps", "8", "--warmup", "2",
           "--no-extras", "--no-cpu", "--nwave", "2501", "--workdir", str(tmp_path / "w"), "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["steps"] == 8
    d = j["scaling_diag"]
    assert d["mode"] == "shard" and d["steps_per_bucket"] == 4 and len(d["per_rank_rt_kernel_ms"]) == 1
    assert d["allgather_send_bytes_per_rank_per_bucket"] == 4 * 10 * 2501 * 8
    # the line certifies its own (gathered) spectra against the oracle, and the replicas comparison of the N > 1
    # line ran: the engine freed and initialised twice more under the live process group
    assert j["parity"]["ok"] and j["parity"]["max_rel_err"] < 1e-9 and j["parity"]["bit_equal_to_plain_launch"]
    assert j["replicas"]["value"] > 0 and len(lines[0]) <= 4096
    # (the line is the contract's record; the diagnostics of the replicas leg are in the detail file it names)
    full = json.load(open(tmp_path / "detail.json"))
    assert full["replicas"]["diag"]["mode"] == "replicas" and full["replicas"]["value"] == pytest.approx(j["replicas"]["value"], rel=1e-5)


CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
from bart_amd import BARTfunc, engine, synthcfg, transit_module as trm

mols = ("H2O", "CO", "CO2", "CH4")
p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
case, cfg = synthcfg.make_worker_case(%(tmp)r, nwave=1777, wnlow=1200.0, opmol=mols, molfit=mols, params=p0, nfilters=5)
w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
rng = np.random.default_rng(3)
pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (6, 7, 9))
pars[..., 3] = np.clip(pars[..., 3], 0, 1)
d_par = torch.from_numpy(pars).cuda()

# (1) the sharded per-step callable with the collective forced: bit-equal to the unsharded step
for s in range(3):
    band0, st0, spec0 = engine.step_batch_dev(d_par[s], w.nfilters, want_spec=True)
    band1, st1, spec1 = engine.step_batch_sharded(d_par[s], w.nfilters, force=True)
    torch.cuda.synchronize()
    assert torch.equal(st0, st1) and torch.equal(spec0, spec1) and torch.equal(band0, band1), s

# (2) GatherPipeline on RCCL's stream: 11 steps in buckets of 4 (two full buckets reuse the double
# buffers, the last one is partly filled), every reassembled spectrum equal to the direct run
n, W = 7, 1777
prof = torch.empty((6, n, engine.nprof()), dtype=torch.float64, device=dev)
stat = torch.empty(n, dtype=torch.int32, device=dev)
import ctypes as C
for s in range(6):
    trm.check(trm.lib().bartrt_step_profiles_dev(C.c_void_p(d_par[s].data_ptr()), n, 9, C.c_void_p(prof[s].data_ptr()),
                                                 C.c_void_p(stat.data_ptr()), engine._stream_ptr()))
direct = [engine.run_batch_dev(prof[s]).clone() for s in range(6)]
pipe = engine.GatherPipeline(n, W, W, 4, dev)
got = []
for i in range(11):
    engine.run_batch_dev(prof[i %% 6], pipe.slot(i))
    done = pipe.submit(i)
    if done is not None:
        got += [done[k].clone() for k in range(done.shape[0])]
for o in pipe.drain(10):
    got += [o[k].clone() for k in range(o.shape[0])]
torch.cuda.synchronize()
assert len(got) == 11
for i in range(11):
    assert torch.equal(got[i], direct[i %% 6]), i

# (3) the uneven-block reassembly on device tensors: three ranks' blocks of a 1777-sample grid
# (592 / 592 / 593) as the all-gather lays them out -- padded, rank-major -- and an even split
for total, world in ((1777, 3), (1776, 3), (10, 4)):
    sizes = engine.block_sizes(total, world)
    full = torch.arange(n * total, dtype=torch.float64, device=dev).view(n, total)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    recv = torch.cat([engine.pad_block(full[:, offs[r]:offs[r + 1]].contiguous(), max(sizes)) for r in range(world)], 0)
    assert recv.shape == (world * n, max(sizes))
    assert torch.equal(engine.reassemble_blocks(recv, n, sizes), full), (total, world)
    # through the collective itself (world 1: the receive buffer is the padded block)
    blk = full[:, :sizes[0]].contiguous()
    out = torch.empty((n, max(sizes)), dtype=torch.float64, device=dev)
    work = dist.all_gather_into_tensor(out, engine.pad_block(blk, max(sizes)), async_op=True)
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(out[:, :sizes[0]], blk)

w.close()
dist.barrier()
dist.destroy_process_group()
print("ok")
"""


def test_sharded_step_and_gather_pipeline_under_nccl_world1(tmp_path):
    code = CHILD % {"root": ROOT, "port": _free_port(), "tmp": str(tmp_path / "case")}
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=900)
    # (RCCL's banner comes out of a C stdio buffer after the script's own last line)
    assert r.returncode == 0 and "ok" in r.stdout.splitlines(), r.stdout[-1500:] + r.stderr[-4000:]
