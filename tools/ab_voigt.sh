# config 5 under the two values of `voigt`: exact Faddeeva per (line, point) against
# profiles tabulated on the width grid; wnosamp 1 and 2160
for v in exact grid; do for o in 1 2160; do
  BARTRT_VOIGT=$v python tools/lbl_bench.py --wnosamp $o 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('voigt $v wnosamp $o ms/spectrum', round(r['seconds_per_spectrum']*1e3,2), 'init_s', round(r['init_s'],2), 'min', r['spectrum_min'], 'max', r['spectrum_max'])"
done; done
