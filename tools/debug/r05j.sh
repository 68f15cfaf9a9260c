for deep in 0 1; do
  BARTRT_QADJ_DEEP=$deep BARTRT_KERNEL=adj16 timeout 200 python tools/ab_small.py 1 2 2>&1 | grep walkers | sed "s/^/pinV deep$deep /"
  AB_CASE=demo BARTRT_QADJ_DEEP=$deep BARTRT_KERNEL=adj16 timeout 200 python tools/ab_small.py 1 3 2>&1 | grep walkers | sed "s/^/pinV deep$deep /"
done
