#!/bin/bash
# Per-kernel times of config 5 (both samplings), on the GPU box from the repo root:
#   [BARTRT_VOIGT_TAYLOR=0] bash tools/lbl_kernel_times.sh <tag>        -> gpurun_out/<tag>/lbl_times.txt
set -u
tag=${1:-try}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag/lblt
rm -rf "$out"; mkdir -p "$out"
cd /tmp
for o in 1 2160; do
  rocprofv3 --kernel-trace --stats -d "$out/o$o" --output-format csv -- python3 "$root/tools/lbl_bench.py" --wnosamp $o > "$out/o$o.json" 2> "$out/o$o.err"
done
cd "$root"
python3 - "$out" <<'PY' | tee "$root/gpurun_out/$tag/lbl_times.txt"
import csv, glob, json, sys
out = sys.argv[1]
for o in (1, 2160):
    line = [l for l in open("%s/o%d.json" % (out, o)) if l.startswith("{")]
    ms = json.loads(line[0])["seconds_per_spectrum"] * 1e3 if line else float("nan")
    fs = glob.glob("%s/o%d/*/*kernel_stats.csv" % (out, o))
    ks = ["%s %.2f" % (r["Name"].split("(")[0].replace("bartrt::", "")[:24], float(r["AverageNs"]) / 1e6)
          for r in list(csv.DictReader(open(fs[0])))[:3]] if fs else []
    print("wnosamp %d: %.2f ms per spectrum | kernels (ms): %s" % (o, ms, " | ".join(ks)))
PY
