mkdir -p gpurun_out/r05zs
for i in $(seq 1 14); do timeout 900 python -m pytest tests/test_gpu_share.py -x -q > gpurun_out/r05zs/run_$i.log 2>&1; tail -1 gpurun_out/r05zs/run_$i.log; if grep -q failed gpurun_out/r05zs/run_$i.log; then cp gpurun_out/r05zs/run_$i.log gpurun_out/r05zs/FAILED_$i.log; else rm gpurun_out/r05zs/run_$i.log; fi; done | tee gpurun_out/r05zs/soak.txt
