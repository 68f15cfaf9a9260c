for integ in 0 1; do for mode in mono_occ mono_ilp; do for n in 10 64 256; do
  st=$((n==10?200:(n==64?60:30)))
  BARTRT_KERNEL=$mode BARTRT_INTEG=$integ python bench.py --walkers $n --steps $st --warmup 10 --no-extras --no-cpu --workdir /tmp/bw 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('INTEG $integ $mode walkers $n', round(r['value']), round(r['ms_per_step'],4), round(r['roofline']['avg_launch_ms'],4), r['roofline']['kernel'])"
done; done; done
