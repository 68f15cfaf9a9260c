mkdir -p gpurun_out/r05e
timeout 900 python tools/debug/qadj_check.py > gpurun_out/r05e/qadj_check.log 2>&1; tail -90 gpurun_out/r05e/qadj_check.log
for k in "" adj8 adj16 octo hexa; do
  BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers | tee -a gpurun_out/r05e/ab_bench.log
done
for k in "" adj8 adj16 r32 hexa; do
  AB_CASE=demo BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 5 2>&1 | grep walkers | tee -a gpurun_out/r05e/ab_demo.log
done
timeout 200 python tools/step_leg.py demo_1walker > gpurun_out/r05e/demo_1walker.json 2>/dev/null; tail -c 900 gpurun_out/r05e/demo_1walker.json; echo
BARTRT_SYNC=stream timeout 200 python tools/step_leg.py demo_1walker > gpurun_out/r05e/demo_1walker_streamsync.json 2>/dev/null; tail -c 900 gpurun_out/r05e/demo_1walker_streamsync.json
