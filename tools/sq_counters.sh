#!/bin/bash
# SQ counters of the eclipse RT kernel at 10 and 256 walkers (separate --pmc passes,
# kernel trace only) -> profiles/<tag>_sq_counters.json with derived figures.
#   bash tools/sq_counters.sh r01m        (on the GPU box, from the repo root)
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/$tag/sq
rm -rf "$out"; mkdir -p "$out"
cd /tmp
B="--steps 60 --warmup 10 --no-cpu --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d "$out/w10" --output-format csv -- python3 "$root/bench.py" $B > "$out/w10.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU -d "$out/w10b" --output-format csv -- python3 "$root/bench.py" $B > "$out/w10b.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d "$out/w256" --output-format csv -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu --no-extras --walkers 256 > "$out/w256.log" 2>&1
cd "$root"
python3 - "$tag" <<'PY'
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
out = {}
for name in ("w10", "w10b", "w256"):
    f = glob.glob("gpurun_out/%s/sq/%s/*/*_counter_collection.csv" % (tag, name))[0]
    acc, dur, kn = collections.defaultdict(list), {}, ""
    for r in csv.DictReader(open(f)):
        if "rt_eclipse" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.setdefault(r["Dispatch_Id"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            kn = r["Kernel_Name"]
    out[name] = {"kernel": kn, "launches": len(dur), "avg_launch_us": sum(dur.values()) / len(dur) / 1e3,
                 "counters_mean_per_launch": {k: sum(v) / len(v) for k, v in sorted(acc.items())}}
for name in ("w10", "w256"):
    c, us = out[name]["counters_mean_per_launch"], out[name]["avg_launch_us"]
    cyc = c["SQ_BUSY_CYCLES"] / 32.0      # summed over the 32 shader engines
    out[name]["derived"] = {
        "shader_clock_GHz": cyc / us / 1e3,
        "fp64_pipe_busy_fraction": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,      # quad-cycles -> cycles, 1024 SIMDs
        "resident_waves_per_simd": c["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc,
        "wave_time": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                      "parked_on_counted_wait": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                      "ready_not_issued": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]}}
path = "gpurun_out/%s/profiles" % tag
os.makedirs(path, exist_ok=True)
for p in (path, "profiles"):
    json.dump(out, open(os.path.join(p, tag + "_sq_counters.json"), "w"), indent=1)
# what bench.py's roofline.bound_measured quotes, tied to the sources it was taken on
sys.path.insert(0, os.getcwd())
import bench
latest = {"source_id": bench.source_id(), "file": tag + "_sq_counters.json",
          "batches": {"10": out["w10"]["derived"], "256": out["w256"]["derived"]}}
for p in (path, "profiles"):
    json.dump(latest, open(os.path.join(p, "sq_latest.json"), "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in out.items()}, indent=1))
PY
