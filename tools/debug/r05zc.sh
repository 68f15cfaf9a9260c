mkdir -p gpurun_out/r05zc
{
for cfg in "BARTRT_MIG=1" "BARTRT_MIG=force"; do
  echo "== $cfg"; env $cfg BARTRT_MIG_DEBUG=1 BARTRT_LIBPATH=bart_amd/libbartrt_mig_dbg.so timeout 300 python tools/ab_small.py 10 2>&1 | grep "walkers\|migration"
done
} | tee gpurun_out/r05zc/probe.txt
