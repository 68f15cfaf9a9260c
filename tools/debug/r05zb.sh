mkdir -p gpurun_out/r05zb
{ ./tools/debug/placement 1570 16384; ./tools/debug/placement 2512 16384; ./tools/debug/placement 1024 16384; } | tee gpurun_out/r05zb/placement.txt
