mkdir -p gpurun_out/r05zl
for k in "" adj16 adj8 hexa octo; do
echo "== BARTRT_KERNEL=$k"; BARTRT_KERNEL=$k timeout 300 python tools/step_leg.py wasp12b_step 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.4g ms_per_step %.4f rt_kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['rt_kernel_ms']))"
done | tee gpurun_out/r05zl/wasp_kernels.txt
