mkdir -p gpurun_out/r05ze
{
for L in bart_amd/libbartrt_twpe2.so bart_amd/libbartrt.so bart_amd/libbartrt_twpe4.so bart_amd/libbartrt_twpe2.so bart_amd/libbartrt.so; do
  echo "== $L"; BARTRT_LIBPATH=$L timeout 300 python tools/transit_bench.py 1 10 64 256 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['walkers'], 'spectra_per_s %d ms_per_step %.4f rt_kernel_ms %.4f' % (d['spectra_per_s'], d['ms_per_step'], d['rt_kernel_ms']))"
done
} 2>&1 | tee gpurun_out/r05ze/transit_ab.txt

