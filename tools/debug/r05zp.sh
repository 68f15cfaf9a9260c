mkdir -p gpurun_out/r05zp
for L in bart_amd/libbartrt.so bart_amd/libbartrt_lbl6.so bart_amd/libbartrt_lbl5.so bart_amd/libbartrt_lbl4.so bart_amd/libbartrt.so bart_amd/libbartrt_lbl6.so; do
  echo "== $L"
  for o in 1 2160; do BARTRT_LIBPATH=$L timeout 300 python tools/lbl_bench.py --wnosamp $o 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('wnosamp $o: %.3f ms per spectrum' % (d['seconds_per_spectrum']*1e3))"; done
done | tee gpurun_out/r05zp/lbl_ab.txt
