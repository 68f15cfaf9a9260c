#!/bin/bash
# Config 5 (line-by-line) evidence, on the GPU box from the repo root:
#   bash tools/profile_lbl.sh r02a
# -> profiles/<tag>_lbl_bench.jsonl (wnosamp 1 and 2160), <tag>_lbl_kernel_stats.csv,
#    <tag>_lbl_o2160_kernel_stats.csv, <tag>_lbl_sq_counters.json
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag/lbl
rm -rf "$out"; mkdir -p "$out"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out/stats_o1" --output-format csv -- python3 "$root/tools/lbl_bench.py" > "$out/o1.json" 2> "$out/o1.err"
rocprofv3 --kernel-trace --stats -d "$out/stats_o2160" --output-format csv -- python3 "$root/tools/lbl_bench.py" --wnosamp 2160 > "$out/o2160.json" 2> "$out/o2160.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d "$out/sq_o1" --output-format csv -- python3 "$root/tools/lbl_bench.py" --reps 2 > "$out/sq_o1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d "$out/sq_o2160" --output-format csv -- python3 "$root/tools/lbl_bench.py" --reps 2 --wnosamp 2160 > "$out/sq_o2160.log" 2>&1
# the width-grid evaluation (cfg `voigt grid`)
BARTRT_VOIGT=grid rocprofv3 --kernel-trace --stats -d "$out/stats_grid_o1" --output-format csv -- python3 "$root/tools/lbl_bench.py" > "$out/grid_o1.json" 2> "$out/grid_o1.err"
BARTRT_VOIGT=grid rocprofv3 --kernel-trace --stats -d "$out/stats_grid_o2160" --output-format csv -- python3 "$root/tools/lbl_bench.py" --wnosamp 2160 > "$out/grid_o2160.json" 2> "$out/grid_o2160.err"
cd "$root"
python3 tools/collect_profiles.py "${tag}_lbl_grid" "$out/stats_grid_o1"
python3 tools/collect_profiles.py "${tag}_lbl_grid_o2160" "$out/stats_grid_o2160"
grep -h '^{' "$out/grid_o1.json" "$out/grid_o2160.json" > "profiles/${tag}_lbl_grid_bench.jsonl"
python3 tools/collect_profiles.py "${tag}_lbl" "$out/stats_o1"
python3 tools/collect_profiles.py "${tag}_lbl_o2160" "$out/stats_o2160"
grep -h '^{' "$out/o1.json" "$out/o2160.json" > "profiles/${tag}_lbl_bench.jsonl"
python3 - "$tag" "$out" <<'PY'
import collections, csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
res = {}
for name in ("sq_o1", "sq_o2160"):
    f = glob.glob("%s/%s/*/*_counter_collection.csv" % (out, name))
    if not f:
        continue
    acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "lbl_" not in k and "rt_eclipse" not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res[name] = {}
    for k in acc:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        us = sum(dur[k].values()) / len(dur[k]) / 1e3
        cyc = c["SQ_BUSY_CYCLES"] / 32.0
        res[name][k] = {"launches": len(dur[k]), "avg_launch_us": us, "counters_mean_per_launch": c,
                        "derived": {"shader_clock_GHz": cyc / us / 1e3,
                                    "valu_pipe_busy_fraction": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                                    "resident_waves_per_simd": c["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc,
                                    "wave_time": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                                  "parked_on_counted_wait": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                                  "ready_not_issued": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]}}}
json.dump(res, open("profiles/%s_lbl_sq_counters.json" % tag, "w"), indent=1)
for n, ks in res.items():
    for k, v in ks.items():
        print(n, k[:50], round(v["avg_launch_us"], 1), "us  valu busy", round(v["derived"]["valu_pipe_busy_fraction"], 3))
PY
mkdir -p "$root/gpurun_out/$tag/profiles" && cp profiles/${tag}_lbl* "$root/gpurun_out/$tag/profiles/" 2>/dev/null
cat "profiles/${tag}_lbl_kernel_stats.csv" | cut -c1-160
