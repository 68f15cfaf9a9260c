#!/bin/bash
# Measurements around the RT kernel (last paragraph), on the GPU box from the
# repo root:   bash tools/misc_bench.sh r02d   -> profiles/<tag>_misc_bench.log
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p "$out"
log="$out/misc_bench.log"
: > "$log"
for t in step_bench worker_step_latency retrieval_rate latency_demo opacity_gen_bench pcie_rate; do
  echo "== $t" >> "$log"
  timeout 600 python3 tools/$t.py 2>/dev/null | grep '^{' >> "$log"
done
cp "$log" "profiles/${tag}_misc_bench.log"
mkdir -p "$out/profiles" && cp "profiles/${tag}_misc_bench.log" "$out/profiles/"
cat "$log" | cut -c1-260
