mkdir -p gpurun_out/r05zn
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05zn/gpu_tests.log 2>&1; tail -3 gpurun_out/r05zn/gpu_tests.log
root=$(pwd); out=$root/gpurun_out/r05zn
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$out/stats_wasp" --output-format csv -- python3 "$root/tools/step_leg.py" wasp12b_step > "$out/stats_wasp.log" 2>&1
cd $root
f=$(find "$out/stats_wasp" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" > profiles/r05_wasp12b_step_kernel_stats.csv
python bench.py > $out/r05_bench.json 2> $out/bench.err
python3 -c "
import json
d=json.load(open('$out/r05_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['max_rel_err']); print(json.dumps(d['configs']['wasp12b_step'])[:420])"
mkdir -p $out/profiles; cp profiles/r05_wasp12b_step_kernel_stats.csv $out/profiles/
find gpurun_out/r05zn -name '*kernel_trace.csv' -delete
