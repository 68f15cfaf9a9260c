import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bart_amd import engine, synth, transit_module as trm
from oracle import rt_oracle as orc
from test_gpu_parity import walkers
import tempfile
tmp = tempfile.mkdtemp()
tag, zero = sys.argv[1], {"top": slice(-9, None), "mid": slice(10, 17)}[sys.argv[1]]
cut, integ = sys.argv[2], int(sys.argv[3])
c = synth.make_case(os.path.join(tmp, "z" + tag), nlayers=30, nwave=130, cia=False)
op = orc.read_opacity(c.opacity)
k = op["kappa"].copy(); k[zero] = 0.0
ids, tg, pr, wnn = op["ids"].copy(), op["temps"].copy(), op["press"].copy(), op["wn"].copy(); del op
synth.write_opacity(c.opacity, ids, tg, pr, wnn, kappa=k)
profs = walkers(c, 2, seed=11)
engine.init(c.tcfg); trm.set_integ(integ); trm.set_cut(cut)
o = orc.OracleEngine(c.tcfg, integ=integ, cut=cut)
ref, got = o.run_batch(profs), engine.run_batch(profs)
engine.walked_begin(); engine.run_batch(profs); print(engine.walked_end()[2])
print("kernel env", os.environ.get("BARTRT_KERNEL"))
np.set_printoptions(linewidth=200)
print("ref ", ref[0, :8], ref[0, 62:68])
print("got ", got[0, :8], got[0, 62:68])
print("relerr max", np.abs(got / ref - 1).max(), "bad lanes", np.where(np.abs(got[0] / ref[0] - 1) > 1e-9)[0][:20])
trm.free_memory()
# per-angle intensities and optical depths of walker 0 (generic kernel outputs)
engine.init(c.tcfg); trm.set_integ(integ); trm.set_cut(cut)
n = trm.get_no_samples()
trm.run_transit(profs[0], n)
A = len(o.angles)
inten = np.zeros((A, n))
trm.check(trm.lib().bartrt_get_intensity(trm._ptr(inten), A, n))
refi = o.intensity(profs[0])
tau, last = engine.get_tau()
_, rtau, rlast = o.run(profs[0], want_tau=True)
for i in (0, 1, 2):
    print("sample", i, "last gpu/orc", last[i], rlast[i])
    print("  tau gpu", tau[i][10:22]); print("  tau orc", rtau[i][10:22]); print("  dtau", (tau[i] - rtau[i])[8:24])
    print("  I gpu", inten[:, i]); print("  I orc", refi[:, i])
trm.free_memory()
