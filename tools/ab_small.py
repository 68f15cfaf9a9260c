"""Launch time of the eclipse RT kernel against batch size for small batches (kernel choice per cut).
usage (GPU box): python tools/ab_small.py [walkers ...]   -- BARTRT_KERNEL / BARTRT_CUT in the environment;
AB_NWAVE=5000 / AB_NLAYERS=50: a shorter grid / fewer layers of the bench shape; AB_CASE=demo: the demo shape (2 501 samples, one molecule)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from bart_amd import engine, synth, transit_module as trm
import tempfile
NW = int(os.environ.get("AB_NWAVE", "10000"))
NL = int(os.environ.get("AB_NLAYERS", "100"))
TD = float(os.environ.get("AB_TEMPDELT", "100"))   # 2600: two temperature planes, shared by every walker (a 64 MB grid)
if os.environ.get("AB_CASE") == "demo":
    NW = 2501
    case = synth.make_case(os.path.join(tempfile.gettempdir(), "bartrt_demo_latency"), nlayers=100, nwave=2501, wnlow=2500.0,
                           opmol=("CH4",), seed=7, reuse=True)
else:
    wd = os.path.join(tempfile.gettempdir(), "bartrt_bench_single_survey8d" + ("" if NW == 10000 else "_%d" % NW) + ("" if NL == 100 else "_L%d" % NL) + ("" if TD == 100 else "_T%d" % TD))
    case = synth.make_case(wd, nlayers=NL, nwave=NW, kappa_model="survey8d", tempdelt=TD, reuse=True)
engine.init(case.tcfg)
for n in [int(x) for x in (sys.argv[1:] or "1 2 3 4 6 8 10".split())]:
    profs = bench.make_profiles(case, n * 8, seed=3).reshape(8, n, -1)
    d = torch.from_numpy(profs).cuda()
    out = torch.empty((n, NW), dtype=torch.float64, device="cuda")
    for i in range(30):
        engine.run_batch_dev(d[i % 8], out)
    torch.cuda.synchronize()
    engine.walked_begin(); engine.run_batch_dev(d[0], out); torch.cuda.synchronize(); kname = engine.walked_end()[2]
    engine.timing_begin(1)
    t0 = time.perf_counter()
    for i in range(200):
        engine.run_batch_dev(d[i % 8], out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    ms, nl = engine.timing_end()
    print("%s cut=%s W=%d walkers %3d: step %.1f us, RT kernel %.1f us  [%s]" % (os.environ.get("BARTRT_KERNEL", "-"), trm.get_cut(), NW, n, dt * 1e6, ms / nl * 1e3, kname), flush=True)
trm.free_memory()
