#!/bin/bash
# SQ counters of the config-5 kernels (one sampling), on the GPU box from the repo root:
#   bash tools/lbl_pmc.sh <tag> <wnosamp>      -> gpurun_out/<tag>/lbl_pmc_<wnosamp>.txt
set -u
tag=${1:-try}; o=${2:-2160}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag/lblpmc_$o
rm -rf "$out"; mkdir -p "$out"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d "$out/p1" --output-format csv -- python3 "$root/tools/lbl_bench.py" --reps 1 --wnosamp $o > "$out/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU -d "$out/p2" --output-format csv -- python3 "$root/tools/lbl_bench.py" --reps 1 --wnosamp $o > "$out/p2.log" 2>&1
cd "$root"
python3 - "$out" <<'PY' | tee "$root/gpurun_out/$tag/lbl_pmc_$o.txt"
import collections, csv, glob, sys
out = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob("%s/%s/*/*_counter_collection.csv" % (out, p))
    if not f:
        print(p, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "lbl_acc" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        print(p, k, {n: "%.4g" % (sum(v) / len(v)) for n, v in acc[k].items()})
PY
