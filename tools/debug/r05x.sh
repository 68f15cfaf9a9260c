mkdir -p gpurun_out/r05x
root=$(pwd)
cd /tmp; export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu --no-extras"
export BARTRT_MIG=0
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r05x/off --output-format csv -- python3 $root/bench.py $B > /dev/null 2>&1
export BARTRT_MIG=1
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r05x/on --output-format csv -- python3 $root/bench.py $B > /dev/null 2>&1
export BARTRT_MIG_LAST=0
timeout 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/r05x/last0 --output-format csv -- python3 $root/bench.py $B > /dev/null 2>&1
cd $root
for d in off on last0; do echo "== $d"; f=$(find gpurun_out/r05x/$d -name '*kernel_stats.csv' | head -1); head -4 "$f" | cut -c1-200; done
find gpurun_out/r05x -name '*kernel_trace.csv' -size +3M -delete
