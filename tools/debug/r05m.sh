mkdir -p gpurun_out/r05m
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r05m/gpu_tests.log 2>&1; tail -6 gpurun_out/r05m/gpu_tests.log
{
  for shape in "--nmol 7 --cia 2" "--nmol 9 --cia 1" "--nmol 4 --cia 2 --angles 12"; do
    BARTRT_RTC_CACHE=/tmp/rtc_cache_m timeout 300 python tools/shape_bench.py $shape 1 2 10 64 2>/dev/null
    BARTRT_RTC=0 timeout 300 python tools/shape_bench.py $shape 1 2 10 64 2>/dev/null
  done
} > gpurun_out/r05m/rtc.txt
cut -c1-330 gpurun_out/r05m/rtc.txt
