mkdir -p gpurun_out/r05zd
timeout 900 python tools/two_groups.py 1 2 3 4 > gpurun_out/r05zd/two_groups.json 2> gpurun_out/r05zd/err.log; cat gpurun_out/r05zd/two_groups.json; tail -3 gpurun_out/r05zd/err.log
