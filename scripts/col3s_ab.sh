#!/bin/bash
# A/B of the pad-on-load inverse with one third of a tile's transform per workgroup (fft_col3.h ColFft3S; MFFT_COL3S=0 / 1), both
# precisions, after the serialised-load fixes of the column-limited c2r and the col3 kernels.  Round 4.
out=gpurun_out/r04_col3s_ab.txt
: > $out
for prec in double single; do
for rep in 1 2 3; do
for m in 0 1; do
  echo "## MFFT_COL3S=$m" >> $out
  MFFT_COL3S=$m python3 scripts/padprof.py 1024 slab $prec >> $out 2>&1
done
done
done
for m in 0 1; do
  echo "## MFFT_COL3S=$m" >> $out
  MFFT_COL3S=$m python3 scripts/padprof.py 512 slab double >> $out 2>&1
  MFFT_COL3S=$m python3 scripts/padprof.py 1024 X double >> $out 2>&1
done
cat $out
