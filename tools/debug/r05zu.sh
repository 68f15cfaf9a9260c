mkdir -p gpurun_out/r05zu
( time python bench.py > gpurun_out/r05zu/r05_bench.json 2> gpurun_out/r05zu/bench.err ) 2>&1 | tail -3
python3 -c "
import json
d=json.load(open('gpurun_out/r05zu/r05_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(list(d['configs'].keys())); print({k:('error' in v) for k,v in d['configs'].items() if isinstance(v,dict)})"
