mkdir -p gpurun_out/r05zg
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05zg/gpu_tests.log 2>&1; tail -3 gpurun_out/r05zg/gpu_tests.log
root=$(pwd); out=$root/gpurun_out/r05zg
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/stats_transit" --output-format csv -- python3 "$root/tools/transit_bench.py" 10 256 > "$out/stats_transit.log" 2>&1
cd $root
python3 tools/collect_profiles.py r05_transit "$out/stats_transit"; cp "$out/stats_transit.log" profiles/r05_transit_bench.jsonl
python bench.py > $out/r05_bench.json 2> $out/bench.err; tail -c 300 $out/r05_bench.json
mkdir -p $out/profiles; cp profiles/r05_transit* $out/profiles/
find gpurun_out/r05zg -name '*kernel_trace.csv' -delete
