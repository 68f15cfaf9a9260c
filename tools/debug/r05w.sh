mkdir -p gpurun_out/r05w
timeout 1200 python -m pytest tests/test_gpu_migration.py -x -q > gpurun_out/r05w/mig_test.log 2>&1; tail -5 gpurun_out/r05w/mig_test.log
B="--steps 200 --warmup 20 --no-cpu --no-extras"
one() { env "$@" timeout 300 python bench.py $B $W 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g  ms_per_step %.4f' % (d['value'], d['ms_per_step']))"; }
{
for cfg in "BARTRT_MIG=0" "BARTRT_MIG=1" "BARTRT_MIG_LAST=0" "BARTRT_MIG_CB=1" "BARTRT_MIG_CB=4" "BARTRT_MIG=0" "BARTRT_MIG=1"; do
  echo "== $cfg"; W=""; one $cfg
done
for w in 12 16 24; do
  for cfg in "BARTRT_MIG=0" "BARTRT_MIG=1" "BARTRT_MIG_LAST=0"; do
    echo "== walkers $w $cfg"; W="--walkers $w"; one $cfg
  done
done
} 2>&1 | tee gpurun_out/r05w/ab.txt
