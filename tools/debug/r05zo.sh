mkdir -p gpurun_out/r05zo
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05zo/gpu_tests.log 2>&1; tail -3 gpurun_out/r05zo/gpu_tests.log
