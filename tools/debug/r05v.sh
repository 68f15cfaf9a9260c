mkdir -p gpurun_out/r05v
timeout 1200 python -m pytest tests/test_gpu_migration.py -x -q > gpurun_out/r05v/mig_test.log 2>&1; tail -25 gpurun_out/r05v/mig_test.log
B="--steps 200 --warmup 20 --no-cpu --no-extras"
for cfg in "BARTRT_MIG=0" "BARTRT_MIG=1" "BARTRT_MIG_CB=2" "BARTRT_MIG_CB=4" "BARTRT_MIG=0" "BARTRT_MIG=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g  ms_per_step %.4f  kernel_us %s frac %s' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_us'), d['roofline'].get('frac')))"
done 2>&1 | tee gpurun_out/r05v/ab.txt
for w in 12 16 24 32; do
  for cfg in "BARTRT_MIG=0" "BARTRT_MIG=1"; do
    echo "== walkers $w $cfg"
    env $cfg timeout 300 python bench.py $B --walkers $w 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g  ms_per_step %.4f  kernel_us %s' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_us')))"
  done
done 2>&1 | tee -a gpurun_out/r05v/ab.txt
