mkdir -p gpurun_out/r05final
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r05final/gpu_tests.log 2>&1; tail -4 gpurun_out/r05final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05final/smoke.log 2>&1; tail -2 gpurun_out/r05final/smoke.log
timeout 3000 bash tools/profile_round.sh r05 > gpurun_out/r05final/profile_round.log 2>&1
timeout 3000 bash tools/profile_r05.sh > gpurun_out/r05final/profile_r05.log 2>&1
mkdir -p gpurun_out/r05final/profiles; cp profiles/r05_* profiles/pmc_latest.json profiles/isa_latest.json gpurun_out/r05final/profiles/ 2>/dev/null
find gpurun_out/r05 gpurun_out/r05x -name '*kernel_trace.csv' -delete 2>/dev/null
find gpurun_out -name '*.csv' -size +2M -delete 2>/dev/null
tail -c 600 profiles/r05_bench.json
