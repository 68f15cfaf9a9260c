mkdir -p gpurun_out/r05g
timeout 900 python tools/debug/qadj_check.py > gpurun_out/r05g/qadj_check.log 2>&1; tail -3 gpurun_out/r05g/qadj_check.log
for deep in 1 0; do
for k in adj8 adj16; do
  BARTRT_QADJ_DEEP=$deep BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers | sed "s/^/deep$deep /" | tee -a gpurun_out/r05g/ab_bench.log
  AB_CASE=demo BARTRT_QADJ_DEEP=$deep BARTRT_KERNEL=$k timeout 200 python tools/ab_small.py 1 2 3 4 5 6 2>&1 | grep walkers | sed "s/^/deep$deep /" | tee -a gpurun_out/r05g/ab_demo.log
done
done
AB_NWAVE=5000 BARTRT_KERNEL=adj16 timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers
AB_NWAVE=5000 BARTRT_KERNEL=adj8 timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers
AB_NWAVE=5000 timeout 200 python tools/ab_small.py 1 2 3 4 2>&1 | grep walkers
