mkdir -p gpurun_out/r05o
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r05o/gpu_tests.log 2>&1; tail -12 gpurun_out/r05o/gpu_tests.log
for f in 1 0; do
  BARTRT_FOLD=$f timeout 200 python tools/step_leg.py demo_1walker 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('fold=$f demo_1walker: run_transit', d['host_call_run_transit_us'], 'step ms', d['ms_per_step'], 'rt kernel ms', d['rt_kernel_ms'], d['kernel'])"
  BARTRT_FOLD=$f timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers | sed "s/^/fold=$f /"
  AB_CASE=demo BARTRT_FOLD=$f timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers | sed "s/^/fold=$f /"
done
timeout 300 python tools/mc3_bench.py 1,2,3 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict): print(k, round(v['aggregate_spectra_per_s']), round(v['us_per_call_median'],1))"
