mkdir -p gpurun_out/r05t
timeout 600 python tools/mc3_bench.py 1,3,10 1500 > gpurun_out/r05t/r05_mc3_service.json 2>/dev/null
{
  for sync in 0 1 2; do
    BARTRT_SVC_SYNC=$sync timeout 300 python tools/mc3_bench.py 10 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); v=d['shared_10']; print('BARTRT_SVC_SYNC=$sync shared_10: %.3e spectra/s, %.1f us per call (loop), %.1f us median call, mean batch %.2f' % (v['aggregate_spectra_per_s'], v['us_per_call_median'], v['call_us_median_of_medians'], v['service']['mean_batch']))"
  done
  BARTRT_SVC_DIRECT_BYTES=0 timeout 300 python tools/mc3_bench.py 10 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); v=d['shared_10']; print('spectra staged through HBM + one DMA copy (BARTRT_SVC_DIRECT_BYTES=0) shared_10: %.3e spectra/s, %.1f us per call' % (v['aggregate_spectra_per_s'], v['us_per_call_median']))"
} > gpurun_out/r05t/r05_mc3_sync_modes.txt
timeout 600 python tools/mc3_bench.py 1,2,3,4,5,6 1500 > gpurun_out/r05t/r05_mc3_few.json 2>/dev/null
python bench.py > gpurun_out/r05t/r05_bench.json 2> gpurun_out/r05t/bench.err
tail -c 400 gpurun_out/r05t/r05_bench.json; cat gpurun_out/r05t/r05_mc3_sync_modes.txt
