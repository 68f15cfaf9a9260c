mkdir -p gpurun_out/r05zi
timeout 600 python tools/mc3_bench.py 1,3,10 1500 > gpurun_out/r05zi/r05_mc3_service.json 2>/dev/null
timeout 600 python tools/mc3_bench.py 1,2,3,4,5,6 1500 > gpurun_out/r05zi/r05_mc3_few.json 2>/dev/null
python bench.py > gpurun_out/r05zi/r05_bench.json 2> gpurun_out/r05zi/bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05zi/r05_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['max_rel_err'])
print(json.dumps(d['configs']['full_step_10'])[:900])
m=json.load(open('gpurun_out/r05zi/r05_mc3_service.json'))
for k,v in m.items():
    if isinstance(v,dict): print(k, round(v['aggregate_spectra_per_s']), round(v['us_per_call_median'],1), round(v['call_us_median_of_medians'],1))
PY
