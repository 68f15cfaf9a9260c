#!/bin/bash
# One rocprofv3 counter pass over bench.py (kernel trace + PMC only), per-kernel means to stdout.
#   bash tools/pmc_pass.sh <name> "<COUNTER ...>" [bench.py arguments]      (on the GPU box, repo root)
# The kernel variant is chosen through the environment (BARTRT_KERNEL=mono ...) of the caller.
set -u
name=$1; counters=$2; shift 2
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/pmc/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
rocprofv3 --kernel-trace --pmc $counters -d "$out" --output-format csv -- python3 "$root/bench.py" --steps 40 --warmup 10 --no-cpu --no-extras "$@" > "$out/run.log" 2>&1
cd "$root"
python3 - "$out" "$name" <<'PY'
import collections, csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
if not f:
    print(sys.argv[2], "NO COUNTER FILE"); print(open(sys.argv[1] + "/run.log").read()[-1500:]); sys.exit(0)
acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "rt_eclipse" not in k:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in acc:
    n = len(dur[k])
    print(json.dumps({"pass": sys.argv[2], "kernel": k, "launches": n, "avg_us": round(sum(dur[k].values()) / n / 1e3, 2),
                      **{c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())}}))
PY
