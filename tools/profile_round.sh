#!/bin/bash
# Runs on the GPU box (from the repo root): the bench line and the rocprofv3
# evidence behind it, into gpurun_out/<tag>/ and profiles/<tag>_* .
#   bash tools/profile_round.sh r01f
set -u
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="--steps 200 --warmup 20 --no-cpu --no-extras"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out/stats_w10" --output-format csv -- python3 "$root/bench.py" $B > "$out/stats_w10.log" 2>&1
# the integration rules side by side: rule 0 and rule 1 (the default) at 10 and 256 walkers
rocprofv3 --kernel-trace --stats -d "$out/stats_w10_i0" --output-format csv -- python3 "$root/bench.py" $B --integ 0 > "$out/stats_w10_i0.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w10_i1" --output-format csv -- python3 "$root/bench.py" $B --integ 1 > "$out/stats_w10_i1.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w256_i0" --output-format csv -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu --no-extras --walkers 256 --integ 0 > "$out/stats_w256_i0.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w256_i1" --output-format csv -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu --no-extras --walkers 256 --integ 1 > "$out/stats_w256_i1.log" 2>&1
# the two cuts side by side and the prefetched form
rocprofv3 --kernel-trace --stats -d "$out/stats_w10_vert" --output-format csv -- python3 "$root/bench.py" $B --cut vertical > "$out/stats_w10_vert.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w256_vert" --output-format csv -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu --no-extras --walkers 256 --cut vertical > "$out/stats_w256_vert.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w10_pf" --output-format csv -- python3 "$root/bench.py" $B --prefetch > "$out/stats_w10_pf.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w10_forest" --output-format csv -- python3 "$root/bench.py" $B --kappa forest > "$out/stats_w10_forest.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w1" --output-format csv -- python3 "$root/bench.py" $B --walkers 1 > "$out/stats_w1.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_w256" --output-format csv -- python3 "$root/bench.py" --steps 20 --warmup 5 --no-cpu --no-extras --walkers 256 > "$out/stats_w256.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" --output-format csv -- python3 "$root/bench.py" --steps 50 --warmup 10 --no-cpu --no-extras > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" --output-format csv -- python3 "$root/bench.py" --steps 50 --warmup 10 --no-cpu --no-extras > "$out/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_calib" --output-format csv -- python3 "$root/tools/pmc_calib.py" > "$out/pmc_calib.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_transit" --output-format csv -- python3 "$root/tools/transit_bench.py" 10 256 > "$out/stats_transit.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_6mol2cia" --output-format csv -- python3 "$root/tools/shape_bench.py" 1 10 256 > "$out/stats_6mol2cia.log" 2>&1
cd "$root"
python3 tools/collect_profiles.py "${tag}_transit" "$out/stats_transit"
python3 tools/collect_profiles.py "${tag}_6mol2cia" "$out/stats_6mol2cia"
cp "$out/stats_transit.log" "profiles/${tag}_transit_bench.jsonl"; cp "$out/stats_6mol2cia.log" "profiles/${tag}_6mol2cia_bench.jsonl"
python3 tools/collect_profiles.py "${tag}_w10" "$out/stats_w10" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_calib"
python3 tools/collect_profiles.py "${tag}_w1" "$out/stats_w1"
for v in w10_i0 w10_i1 w256_i0 w256_i1 w10_vert w256_vert w10_pf w10_forest; do python3 tools/collect_profiles.py "${tag}_$v" "$out/stats_$v"; done
# L2 hits / misses of the RT launch (the figures DESIGN.md section 6 quotes)
bash tools/pmc_pass.sh tcc "TCC_HIT_sum TCC_MISS_sum" > "profiles/${tag}_tcc.jsonl" 2>&1
grep -q '"kernel"' "profiles/${tag}_tcc.jsonl" || bash tools/pmc_pass.sh tcc "TCC_HIT TCC_MISS" > "profiles/${tag}_tcc.jsonl" 2>&1
python3 tools/collect_profiles.py "${tag}_w256" "$out/stats_w256"
# instruction mix of the build (the bench line's fp64 figure is computed from it)
python3 tools/isa_stats.py > "profiles/${tag}_isa_rt_eclipse_fast_5_4_1_sq.txt"
python3 tools/isa_stats.py --ilp > "profiles/${tag}_isa_rt_eclipse_fast_5_4_1_sq_ilp.txt"
# (the default rule's kernel: its figures feed the bench line's fp64 record)
python3 tools/isa_stats.py --ilp rt_eclipse_simpsonILi5ELi4ELi2ELb1ELi1ELb0E > "profiles/${tag}_isa_rt_eclipse_simpson_5_4_2_sq_ilp.txt"
python3 tools/isa_stats.py --json profiles/isa_latest.json rt_eclipse_simpson_slantILi5ELi4ELi2ELb1ELi1ELb0E > "profiles/${tag}_isa_rt_eclipse_simpson_slant_5_4_2_sq_ilp.txt"
# SQ pass of the same build (roofline.bound_measured)
bash tools/sq_counters.sh "$tag" > "$out/sq.log" 2>&1
# the bench line last, so that its `traffic` / `bound_measured` / `fp64` are this build's figures
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
cp "$out/bench.json" "profiles/${tag}_bench.json"
cp bench_detail.json "profiles/${tag}_bench_detail.json"
mkdir -p "$out/profiles" && cp profiles/${tag}_* profiles/pmc_latest.json profiles/isa_latest.json profiles/sq_latest.json "$out/profiles/" 2>/dev/null
tail -c 1800 "$out/bench.json"
# the round's evidence must be the shipped library's: every profiler record carries bartrt_build_id(), and the line
# quotes them -- a record taken on another build fails the script
python3 - "$tag" <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
sid = bench.source_id()
bad = [f for f in ("pmc_latest.json", "isa_latest.json", "sq_latest.json")
       if json.load(open("profiles/" + f)).get("source_id") != sid]
line = json.loads(open("profiles/%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = line["roofline"]
missing = [k for k in ("traffic", "traffic_source", "bound_measured", "fp64_frac") if r.get(k) is None]
print("library id %s; line id %s; stale records: %s; null in the line: %s" % (sid, line.get("source_id"), bad, missing))
sys.exit(1 if (bad or missing or line.get("source_id") != sid or len(open("profiles/%s_bench.json" % sys.argv[1]).read()) > 4096) else 0)
PY
