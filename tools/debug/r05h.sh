mkdir -p gpurun_out/r05h
for k in "" adj8 adj16; do
  AB_TEMPDELT=2600 BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 2>&1 | grep walkers | sed "s/^/Nt2 /" | tee -a gpurun_out/r05h/nt2.log
done
timeout 1500 python -m pytest tests/test_gpu_rtc.py -x -q > gpurun_out/r05h/rtc_test.log 2>&1; tail -25 gpurun_out/r05h/rtc_test.log
