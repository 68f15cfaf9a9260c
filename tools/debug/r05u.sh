mkdir -p gpurun_out/r05u
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05u/gpu_tests.log 2>&1; tail -4 gpurun_out/r05u/gpu_tests.log
for i in 1 2 3; do timeout 300 python tools/step_leg.py full_step_10 2>/dev/null | tail -1; done > gpurun_out/r05u/full_step.txt
cat gpurun_out/r05u/full_step.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05u/prof -o fs -- python3 $GRAFT_REPO_ROOT/tools/step_leg.py full_step_10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r05u/prof -name '*kernel_stats.csv' | head -1); head -12 "$f" | cut -c1-220
python tools/ab_small.py 2>/dev/null | tail -12
