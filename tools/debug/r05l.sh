mkdir -p gpurun_out/r05l
timeout 900 python -m pytest tests/test_gpu_share.py -x -q > gpurun_out/r05l/share.log 2>&1; tail -5 gpurun_out/r05l/share.log
for sm in 4 0; do
BARTRT_SVC_SPLIT_MAX=$sm timeout 400 python tools/mc3_bench.py 2,3,4 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict): print('split_max=$sm', k, round(v['aggregate_spectra_per_s']), round(v['us_per_call_median'],1), v.get('service',{}).get('mean_batch'))" | tee -a gpurun_out/r05l/few.log
done
