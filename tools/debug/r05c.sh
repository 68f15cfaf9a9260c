mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_cia_layouts.py tests/test_gpu_share.py -x -q -s > gpurun_out/r05c/tests.log 2>&1; tail -15 gpurun_out/r05c/tests.log
for leg in full_step_10 wasp12b_step wasp12b_shard8; do
  timeout 300 python tools/step_leg.py $leg > gpurun_out/r05c/$leg.json 2> gpurun_out/r05c/$leg.err; tail -c 1500 gpurun_out/r05c/$leg.json; echo
done
R=$(pwd)
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05c/stats_full_step --output-format csv -- python3 $R/tools/step_leg.py full_step_10 > $R/gpurun_out/r05c/stats_full_step.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05c/stats_wasp_step --output-format csv -- python3 $R/tools/step_leg.py wasp12b_step > $R/gpurun_out/r05c/stats_wasp_step.log 2>&1
cd $R
find gpurun_out/r05c -name "*kernel_stats.csv" | while read f; do echo $f; head -8 $f | cut -c1-200; done
find gpurun_out/r05c -name "*_kernel_trace.csv" -delete; find gpurun_out/r05c -name "*.db" -delete
