mkdir -p gpurun_out/r05za
{
for cfg in "BARTRT_MIG=0" "BARTRT_MIG=1" "BARTRT_LIBPATH=bart_amd/libbartrt_mig_inline.so" "BARTRT_MIG=force"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/ab_small.py 10 16 2>&1 | grep walkers
done
} | tee gpurun_out/r05za/probe.txt
