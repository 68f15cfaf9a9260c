mkdir -p gpurun_out/r05zm
for k in "" adj16; do
echo "== BARTRT_KERNEL=$k"; BARTRT_KERNEL=$k AB_NWAVE=2424 timeout 300 python tools/ab_small.py 8 9 10 11 12 13 2>&1 | grep walkers | cut -c1-140
done | tee gpurun_out/r05zm/cols.txt
