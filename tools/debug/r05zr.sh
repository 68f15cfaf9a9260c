mkdir -p gpurun_out/r05zr
for w in 6 7 10 12 13 14 19 20 26 27; do python bench.py --steps 200 --warmup 20 --no-cpu --no-extras --walkers $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walkers $w: %.4g spectra/s, %.1f us per step, %.2f us per spectrum' % (d['value'], d['ms_per_step']*1e3, d['ms_per_step']*1e3/$w))"; done | tee gpurun_out/r05zr/walkers.txt
