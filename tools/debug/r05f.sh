mkdir -p gpurun_out/r05f
for lib in "" qilp; do
for k in adj8 adj16; do
  L=""; [ -n "$lib" ] && L=$(pwd)/bart_amd/libbartrt_$lib.so
  BARTRT_LIBPATH=$L BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 2>&1 | grep walkers | sed "s/^/$lib /" | tee -a gpurun_out/r05f/ab_bench.log
  AB_CASE=demo BARTRT_LIBPATH=$L BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 5 2>&1 | grep walkers | sed "s/^/$lib /" | tee -a gpurun_out/r05f/ab_demo.log
done
done
