set -x
python -m pytest tests/test_gpu_parity.py tests/test_oracle_kat.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -15
for integ in 0 1; do for n in 10 256; do
  st=$((n==10?300:30))
  BARTRT_INTEG=$integ python bench.py --walkers $n --steps $st --warmup 10 --no-extras --no-cpu --workdir /tmp/bw 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('INTEG $integ walkers $n', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['kernel'])"
done; done
