mkdir -p gpurun_out/r05zh
timeout 300 python tools/step_leg.py full_step_10 2>/dev/null | tail -1 > gpurun_out/r05zh/full_step_10.json; python3 -c "
import json
d=json.load(open('gpurun_out/r05zh/full_step_10.json')); h=d['host_call_step_batch']; print('median %.1f mean %.1f p90 %.1f' % (h['median_us'], h['us_per_step'], h['p90_us']), h['slowest_calls']); print(d['synchronised_every_step']['us_per_step'], d['queued_back_to_back']['us_per_step'])"
