mkdir -p gpurun_out/r05d
BARTRT_LIBPATH=$(pwd)/bart_amd/libbartrt_phase.so timeout 300 python tools/debug/phase_clock.py > gpurun_out/r05d/phase.log 2>&1; cat gpurun_out/r05d/phase.log | tail -30
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05d/gpu_tests.log 2>&1; tail -8 gpurun_out/r05d/gpu_tests.log
