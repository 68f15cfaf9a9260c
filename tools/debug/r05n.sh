mkdir -p gpurun_out/r05n
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_worker_golden.py tests/test_gpu_worker_fuzz.py tests/test_worker.py tests/test_sampler.py -x -q -m gpu > gpurun_out/r05n/tests.log 2>&1; tail -25 gpurun_out/r05n/tests.log
for f in 1 0; do
  BARTRT_BAND_FUSE=$f timeout 300 python tools/step_leg.py full_step_10 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('fuse=$f full_step_10', {k:{kk:round(vv,1) for kk,vv in v.items()} for k,v in d.items() if isinstance(v,dict)}, d.get('band_fluxes_bit_stable'))"
  BARTRT_BAND_FUSE=$f timeout 300 python tools/step_leg.py wasp12b_step 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('fuse=$f wasp12b_step', round(d['ms_per_step']*1e3,1), 'us per step, RT', round(d['rt_kernel_ms']*1e3,1), round(d['value']))"
done
