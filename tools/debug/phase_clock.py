"""Phase clock of step_prep_profiles / prep_profiles (development build: python tools/ab_build.py phase step.hip
-DBARTRT_PHASE_CLOCK; run with BARTRT_LIBPATH=bart_amd/libbartrt_phase.so).  Prints the 100 MHz stamps of workgroup 0
relative to the kernel's first, for the whole step at the headline shape and at the WASP shape."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bart_amd import BARTfunc, engine, synthcfg, transit_module as trm
mols = ("H2O", "CO", "CO2", "CH4")
p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
names = {0: "entry", 1: "staged (barrier)", 2: "pow() of kappa, gamma, factors (barrier)", 3: "T(p) raw (barrier)", 4: "profile written (barrier)",
         8: "prep: entry barrier", 9: "prep: H terms (barrier)", 10: "prep: serial radii (barrier)", 11: "prep: records written"}
for nwave, nf in ((2424, 4), (10000, 10)):
    d = os.path.join(tempfile.gettempdir(), "bartrt_cfg_phase_%d" % nwave)
    case, cfg = synthcfg.make_worker_case(d, nwave=nwave, wnlow=910.0, opmol=mols, molfit=mols, params=p0, nfilters=nf, reuse=True)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    rng = np.random.default_rng(5)
    pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (10, 9)); pars[:, 3] = np.clip(pars[:, 3], 0, 1)
    dp = torch.from_numpy(pars).cuda()
    res = []
    for it in range(30):
        engine.step_batch_dev(dp, w.nfilters); torch.cuda.synchronize()
        out = (C.c_ulonglong * 32)()
        assert trm.lib().bartrt_debug_phase_clock(out, 32) == 0
        res.append(np.array(out[:], dtype=np.int64))
    r = np.median(np.array(res[5:]), axis=0)
    print("W = %d" % nwave)
    for k in sorted(names):
        print("   %-45s +%.2f us" % (names[k], (r[k] - r[0]) / 100.0))
    w.close()
