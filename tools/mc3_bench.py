"""The N-worker-process leg on its own (tools/bench_configs.py: mc3_processes): python tools/mc3_bench.py [nprocs,..] [steps]
-> one JSON line.  Environment (BARTRT_SVC_*) reaches the workers."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if __name__ == "__main__":
    import bench_configs
    nprocs = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 3, 10)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    d = os.environ.get("BARTRT_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "bartrt_bench_headline")
    print(json.dumps(bench_configs.mc3_processes(d, "survey8d", nprocs=nprocs, steps=steps)))
