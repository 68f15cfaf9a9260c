"""Two real ranks on the real kernels: two processes, both on device 0, each an engine on ITS
wavenumber block (`--shard r 2`), the sharded per-step callable and an eleven-step GatherPipeline against the unsharded
run.  RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the process group is tried as nccl first and
falls back to gloo -- engine.allgather_blocks then stages the blocks through pinned host memory behind the same
interface.  It is the first time the compute stream / collective ordering, the block sizes of an odd grid (1777 =
888 + 889) and the reassembly meet a second rank; the 8-GPU form stays the driver's to run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
rank = int(sys.argv[1]); world = 2
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
backend = sys.argv[2]
kw = {"device_id": dev} if backend == "nccl" else {}
dist.init_process_group(backend, init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world, **kw)
if backend == "nccl":
    # two ranks, one device: RCCL says no at its first collective
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
from bart_amd import BARTfunc, engine, synthcfg, transit_module as trm

mols = ("H2O", "CO", "CO2", "CH4")
p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
case, cfg = synthcfg.make_worker_case(%(tmp)r + "/r%%d" %% rank, nwave=1777, wnlow=1200.0, opmol=mols, molfit=mols, params=p0, nfilters=5)
ref = np.load(%(tmp)r + "/ref.npz")
pars = ref["pars"]
kernel_by = sys.argv[3]
os.environ["BARTRT_KERNEL_BY"] = kernel_by
w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg), shard=(rank, world))
lo, hi = engine.local_range()
assert (lo, hi) == ((0, 888) if rank == 0 else (888, 1777)), (lo, hi)
assert trm.get_kernel_by() == kernel_by
d_par = torch.from_numpy(pars).cuda()
exact = kernel_by == "whole"

def same(a, b, what):
    a, b = a.cpu().numpy(), np.asarray(b)
    if exact:
        assert np.array_equal(a, b), what
    else:
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13 * np.abs(b).max(), err_msg=str(what))

# (1) the sharded per-step callable: every rank ends with the unsharded step's full spectra, band fluxes, statuses
for s in range(3):
    band, st, spec = engine.step_batch_sharded(d_par[s], w.nfilters)
    torch.cuda.synchronize()
    assert spec.shape == (7, 1777)
    same(spec, ref["spec"][s], ("spec", s)); same(band, ref["band"][s], ("band", s))
    assert np.array_equal(st.cpu().numpy(), ref["status"][s]), s

# (2) GatherPipeline: 11 steps in buckets of 4 over the two ranks' blocks
import ctypes as C
n, W = 7, 1777
prof = torch.empty((6, n, engine.nprof()), dtype=torch.float64, device=dev)
stat = torch.empty(n, dtype=torch.int32, device=dev)
for s in range(6):
    trm.check(trm.lib().bartrt_step_profiles_dev(C.c_void_p(d_par[s].data_ptr()), n, 9, C.c_void_p(prof[s].data_ptr()),
                                                 C.c_void_p(stat.data_ptr()), engine._stream_ptr()))
pipe = engine.GatherPipeline(n, hi - lo, W, 4, dev)
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
    same(got[i], ref["spec"][i %% 6], ("pipeline", i))
w.close()
dist.barrier()
dist.destroy_process_group()
print("ok " + json.dumps({"rank": rank, "backend": backend, "kernel_by": kernel_by}))
"""

REF = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
from bart_amd import BARTfunc, engine, synthcfg
mols = ("H2O", "CO", "CO2", "CH4")
p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
case, cfg = synthcfg.make_worker_case(%(tmp)r + "/ref", nwave=1777, wnlow=1200.0, opmol=mols, molfit=mols, params=p0, nfilters=5)
w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
rng = np.random.default_rng(3)
pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (6, 7, 9))
pars[..., 3] = np.clip(pars[..., 3], 0, 1)
d_par = torch.from_numpy(pars).cuda()
spec, band, status = [], [], []
for s in range(6):
    b, st, sp = engine.step_batch_dev(d_par[s], w.nfilters, want_spec=True)
    torch.cuda.synchronize()
    spec.append(sp.cpu().numpy()); band.append(b.cpu().numpy()); status.append(st.cpu().numpy())
np.savez(%(tmp)r + "/ref.npz", pars=pars, spec=np.array(spec), band=np.array(band), status=np.array(status))
w.close()
print("ok")
"""


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "1")
    return e


def _pair(tmp, backend, kernel_by, timeout=600):
    code = CHILD % {"root": ROOT, "port": _free_port(), "tmp": tmp}
    ps = [subprocess.Popen([sys.executable, "-c", code, str(r), backend, kernel_by], env=_env(), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in ps:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            o, e = p.communicate()
        outs.append((p.returncode, o, e))
    return outs


@pytest.fixture(scope="module")
def reference(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("two"))
    r = subprocess.run([sys.executable, "-c", REF % {"root": ROOT, "tmp": tmp}], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]
    return tmp


@pytest.fixture(scope="module")
def backend(reference):
    """nccl if RCCL takes two ranks on one device, else gloo (recorded in the test's output)."""
    outs = _pair(reference, "nccl", "whole", timeout=240)
    if all(rc == 0 and any(l.startswith("ok ") for l in o.splitlines()) for rc, o, _ in outs):
        return "nccl"
    return "gloo"


@pytest.mark.parametrize("kernel_by", ["whole", "local"])
def test_two_ranks_on_one_gpu_against_the_unsharded_run(reference, backend, kernel_by):
    outs = _pair(reference, backend, kernel_by)
    for rc, o, e in outs:
        assert rc == 0 and any(l.startswith("ok ") for l in o.splitlines()), o[-1500:] + e[-4000:]
    rep = [json.loads([l for l in o.splitlines() if l.startswith("ok ")][0][3:]) for _, o, _ in outs]
    assert sorted(r["rank"] for r in rep) == [0, 1] and all(r["backend"] == backend for r in rep)
    print("two ranks on one GPU ran under", backend)
