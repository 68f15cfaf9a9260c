timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
timeout 400 python tools/mc3_bench.py 3,10 1500 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict): print(k, round(v['aggregate_spectra_per_s']), 'loop', round(v['us_per_call_median'],1), 'median call', round(v['call_us_median_of_medians'],1), 'init', round(v['init_s_median'],3))"
