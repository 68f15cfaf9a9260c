# ray-grid sizes: the size's own single-wave kernel against the generic one, per (layer, wavenumber, angle)
for A in 1 3 4 5 6 9; do for integ in 0 1; do for k in default generic; do
  if [ $k = generic ]; then export BARTRT_KERNEL=generic; else unset BARTRT_KERNEL; fi
  python tools/shape_bench.py --nmol 4 --cia 1 --angles $A --integ $integ 1 10 256 2>/dev/null
done; done; done
