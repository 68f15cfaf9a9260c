"""A/B timing of the RT kernel variants on the bench grid.

usage: python tools/ab_kernels.py [walkers ...]
Each (variant, batch) runs bench.py in a child process (the variant switches are
read once per process) and reports step time and RT-kernel time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ("default", {}),
    ("mono", {"BARTRT_KERNEL": "mono"}),
    ("split", {"BARTRT_KERNEL": "split"}),
    ("quad", {"BARTRT_KERNEL": "quad"}),
]


only = [v for v in os.environ.get("AB_ONLY", "").split(",") if v]
if only:
    VARIANTS = [v for v in VARIANTS if v[0] in only]


def main():
    batches = [int(a) for a in sys.argv[1:]] or [1, 4, 10, 16, 32, 64, 256]
    wd = os.path.join(os.environ.get("TMPDIR", "/tmp"), "bartrt_ab")
    rows = []
    for n in batches:
        for name, env in VARIANTS:
            steps = max(10, min(300, 3000 // n))
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--walkers", str(n), "--steps", str(steps),
                   "--warmup", "10", "--no-extras", "--no-cpu", "--workdir", wd]
            out = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(name, n, "FAILED", out.stderr[-500:])
                continue
            r = json.loads(line[-1])
            rows.append({"variant": name, "walkers": n, "spectra_per_s": round(r["value"]),
                         "ms_per_step": round(r["ms_per_step"], 4),
                         "rt_kernel_ms": round(r["roofline"]["avg_launch_ms"], 4)})
            print(json.dumps(rows[-1]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "ab_kernels.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
