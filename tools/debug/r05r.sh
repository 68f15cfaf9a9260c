cat > /tmp/run3.py <<'PY'
import json, os, subprocess, sys, tempfile
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
from bart_amd import synth
d = os.path.join(tempfile.gettempdir(), "bartrt_bench_headline")
case = synth.make_case(d, nlayers=100, nwave=10000, kappa_model="survey8d", reuse=True)
cfg = case.tcfg + ".share"
open(cfg, "w").write(open(case.tcfg).read().rstrip("\n") + "\nshareOpacity\n")
n = int(sys.argv[1])
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
ps = [subprocess.Popen([sys.executable, "tools/mc3_child.py", cfg, str(r), "3000"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(n)]
def expect(p, w):
    for line in p.stdout:
        if line.startswith(w + " "): return json.loads(line[len(w) + 1:])
    raise RuntimeError(p.stderr.read()[-2000:])
[expect(p, "ready") for p in ps]
for p in ps: p.stdin.write("go\n"); p.stdin.flush()
done = [expect(p, "done") for p in ps]
for p in ps: p.stdin.write("bye\n"); p.stdin.flush()
for p in ps: p.wait(timeout=120)
for d in done:
    print(n, "rank", d["rank"], "mean", round(d["us_per_step"], 1), "median", round(d["call_us_median"], 1), "max", round(d["call_us_max"]), "over2x", d["calls_over_twice_the_median"], "slowest", d["slowest_calls"])
PY
python /tmp/run3.py 3
python /tmp/run3.py 10 | head -4
BARTRT_SVC_SPIN_US=100000 python /tmp/run3.py 3
