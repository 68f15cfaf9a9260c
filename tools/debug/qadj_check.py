"""rt_eclipse_qadj (rows on adjacent lanes) against the oracle and against rt_eclipse_quad<ALLR>, and its launch time.
GPU box: python tools/debug/qadj_check.py"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

CHILD = r"""
import sys, json, os
sys.path.insert(0, %(root)r)
import numpy as np
from bart_amd import engine, synth, transit_module as trm
sys.path.insert(0, os.path.join(%(root)r, "tests"))
out = {}
for name, kw in %(cases)r:
    c = synth.make_case(os.path.join(%(tmp)r, name), **kw)
    engine.init(c.tcfg)
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4):
        profs = []
        for k in range(n):
            t = np.clip(c.temp0 * (0.8 + 0.4 * rng.random()) + 150 * rng.normal(size=c.temp0.shape) * 0.2, 420, 2950)
            profs.append(c.profiles(temp=t).ravel())
        profs = np.array(profs)
        engine.walked_begin()
        a = engine.run_batch(profs)
        kname = engine.walked_end()[2]
        trm.set_cloudtop(-1.0)
        b = engine.run_batch(profs)
        trm.free_memory(); engine.init(c.tcfg)
        np.save(os.path.join(%(tmp)r, "%%s_%%d_%%s.npy" %% (name, n, %(mode)r)), np.array([a, b]))
        np.save(os.path.join(%(tmp)r, "%%s_%%d_prof.npy" %% (name, n)), profs)
        out["%%s_%%d" %% (name, n)] = kname
    trm.free_memory()
print("RES " + json.dumps(out))
"""

def main():
    tmp = tempfile.mkdtemp(prefix="qadj_")
    cases = [("demo", dict(nlayers=100, nwave=2501, wnlow=2500.0, opmol=("CH4",), seed=7)),
             ("bench", dict(nlayers=100, nwave=4000, kappa_model="survey8d")),
             ("cut", dict(nlayers=61, nwave=1500, toomuch=1.5)),
             ("lin", dict(nlayers=100, nwave=3000, extra_keys={"cia_interp": "linear"}))]
    names = {}
    for mode in ("", "adj8", "adj16", "octo", "hexa"):
        env = dict(os.environ)
        env.pop("BARTRT_KERNEL", None)
        if mode:
            env["BARTRT_KERNEL"] = mode
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "cases": cases, "tmp": tmp, "mode": mode or "default"}],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        names[mode or "default"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RES ")][0][4:])
    from oracle import rt_oracle as orc
    from bart_amd import synth
    worst = {}
    for name, kw in cases:
        c = synth.make_case(os.path.join(tmp, name), write=False, **kw)
        o = orc.OracleEngine(c.tcfg)
        for n in (1, 2, 3, 4):
            profs = np.load(os.path.join(tmp, "%s_%d_prof.npy" % (name, n)))
            ref = [o.run_batch(profs)]
            o.set_cloudtop(-1.0); ref.append(o.run_batch(profs)); o.c.has_cloud = 0
            ref = np.array(ref)
            for mode in ("default", "adj8", "adj16", "octo", "hexa"):
                got = np.load(os.path.join(tmp, "%s_%d_%s.npy" % (name, n, mode)))
                err = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12 * np.abs(ref).max())))
                worst[(name, n, mode)] = err
                print("%-6s n=%d %-8s %-55s max rel err vs oracle %.2e" % (name, n, mode, names[mode]["%s_%d" % (name, n)], err), flush=True)
    bad = {k: v for k, v in worst.items() if not v < 1e-10}
    print("WORST", max(worst.values()), "BAD", bad)

if __name__ == "__main__":
    main()
