mkdir -p gpurun_out/r05k
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r05k/gpu_tests.log 2>&1; tail -15 gpurun_out/r05k/gpu_tests.log
