mkdir -p gpurun_out/r05b
timeout 600 python -m pytest tests/test_gpu_share.py -x -q > gpurun_out/r05b/share_tests.log 2>&1; tail -5 gpurun_out/r05b/share_tests.log
for sync in 0 1 2; do
  BARTRT_SVC_SYNC=$sync timeout 300 python tools/mc3_bench.py 10 1500 > gpurun_out/r05b/mc3_sync$sync.json 2> gpurun_out/r05b/mc3_sync$sync.err
  python - <<PY
import json; d=json.load(open("gpurun_out/r05b/mc3_sync$sync.json")); print("sync$sync", {k:(v["aggregate_spectra_per_s"], v["us_per_call_median"], v["call_us_median_of_medians"]) for k,v in d.items() if isinstance(v, dict)})
PY
done
BARTRT_SVC_SYNC=1 BARTRT_SVC_DIRECT_BYTES=0 timeout 300 python tools/mc3_bench.py 10 1500 > gpurun_out/r05b/mc3_staged.json 2>/dev/null
python - <<PY
import json; d=json.load(open("gpurun_out/r05b/mc3_staged.json")); print("staged", {k:(v["aggregate_spectra_per_s"], v["us_per_call_median"]) for k,v in d.items() if isinstance(v, dict)})
PY
BARTRT_SVC_SYNC=1 timeout 300 python tools/mc3_bench.py 1,2,3,4 1500 > gpurun_out/r05b/mc3_few.json 2>/dev/null
python - <<PY
import json; d=json.load(open("gpurun_out/r05b/mc3_few.json")); print("few", {k:(v["aggregate_spectra_per_s"], v["us_per_call_median"]) for k,v in d.items() if isinstance(v, dict)})
PY
