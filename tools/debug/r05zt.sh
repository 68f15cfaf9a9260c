mkdir -p gpurun_out/r05zt
for i in 1 2 3 4; do timeout 1500 python -m pytest tests/ -q -m gpu > gpurun_out/r05zt/run_$i.log 2>&1; tail -1 gpurun_out/r05zt/run_$i.log; grep -q "failed" gpurun_out/r05zt/run_$i.log || rm gpurun_out/r05zt/run_$i.log; done | tee gpurun_out/r05zt/soak.txt
