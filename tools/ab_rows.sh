#!/bin/bash
# Rows per step of the all-rays layer-parallel kernel (rt_eclipse_quad<..., ALLR>), small batches (GPU box).
for k in octo hexa r32 octorays quadrays; do
  AB_CASE=demo BARTRT_KERNEL=$k python3 tools/ab_small.py 1 2 3 4 5
done
for k in octo hexa r32 mono_ilp; do
  AB_NWAVE=5000 BARTRT_KERNEL=$k python3 tools/ab_small.py 1 2 3
done
python3 tools/ab_small.py 1 2
python3 tools/latency_demo.py
