mkdir -p gpurun_out/r05zf
{
for L in bart_amd/libbartrt.so bart_amd/libbartrt_qadj3.so bart_amd/libbartrt_quad3.so bart_amd/libbartrt.so; do
  echo "== $L"
  BARTRT_LIBPATH=$L timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers
  BARTRT_LIBPATH=$L AB_NWAVE=5000 timeout 200 python tools/ab_small.py 1 2 4 2>&1 | grep walkers
  BARTRT_LIBPATH=$L AB_CASE=demo timeout 200 python tools/ab_small.py 1 2 3 5 6 2>&1 | grep walkers
done
} | tee gpurun_out/r05zf/few_ab.txt
