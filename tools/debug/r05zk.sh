mkdir -p gpurun_out/r05zk
for L in bart_amd/libbartrt_band_old.so bart_amd/libbartrt.so bart_amd/libbartrt_band_old.so bart_amd/libbartrt.so; do
echo "== $L"; BARTRT_LIBPATH=$L timeout 300 python tools/step_leg.py full_step_10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('queued %.1f sync %.1f host median %.1f mean %.1f' % (d['queued_back_to_back']['us_per_step'], d['synchronised_every_step']['us_per_step'], d['host_call_step_batch']['median_us'], d['host_call_step_batch']['us_per_step']))"
done | tee gpurun_out/r05zk/band_ab.txt
timeout 900 python -m pytest tests/ -x -q -m gpu -k "band or step or worker or energy" 2>&1 | tail -2
