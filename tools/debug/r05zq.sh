mkdir -p gpurun_out/r05zq
for k in "" adj16 adj8 hexa octo mono_ilp; do
echo "== BARTRT_KERNEL=$k"; BARTRT_KERNEL=$k AB_NWAVE=1250 timeout 300 python tools/ab_small.py 8 10 12 2>&1 | grep walkers | cut -c1-150
done | tee gpurun_out/r05zq/shard1250.txt
