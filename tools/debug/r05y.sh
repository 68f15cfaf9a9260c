mkdir -p gpurun_out/r05y
B="--steps 200 --warmup 20 --no-cpu --no-extras"
one() { env "$@" timeout 300 python bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.4g  ms_per_step %.4f' % (d['value'], d['ms_per_step']))"; }
{
echo "== off"; one BARTRT_MIG=0
for v in "" _inline _noend _noendplain; do
  L=bart_amd/libbartrt${v:+_mig}$v.so
  echo "== $L on"; one BARTRT_LIBPATH=$L
  echo "== $L last0"; one BARTRT_LIBPATH=$L BARTRT_MIG_LAST=0
done
} 2>&1 | tee gpurun_out/r05y/ab.txt
