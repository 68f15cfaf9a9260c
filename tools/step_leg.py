"""One leg of tools/bench_configs.py on its own (for rocprofv3 --kernel-trace --stats: the per-kernel split of a step):
python tools/step_leg.py full_step_10|wasp12b_step|wasp12b_shard8|demo_1walker [integ]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if __name__ == "__main__":
    import bench_configs
    print(json.dumps(getattr(bench_configs, sys.argv[1])(int(sys.argv[2]) if len(sys.argv) > 2 else 1)))
