#!/bin/bash
# c2r mirrors through LDS on the 20-values plans (real 800 / 1000 / 1600 / 2000 / 4000): plain rows and the column-limited rows of the
# pruned 2/3-rule, MFFT_C2R_MLDS=0 (second load) against 2 (wherever built)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/mlds_e20.txt
: > $O
for n in 800 1000 1600; do
  for p in double single; do
    for m in 0 2; do
      echo "## $n^3 $p MFFT_C2R_MLDS=$m" | tee -a $O
      MFFT_C2R_MLDS=$m timeout 600 python scripts/maskprof.py $n $p slab 2>&1 | tee -a $O
      MFFT_C2R_MLDS=$m timeout 600 python scripts/meshprof.py $n $n $n $p 2>&1 | grep "bwd_z" | tee -a $O
    done
  done
done
for n in 2000 4000; do
  for p in double single; do
    for m in 0 2; do
      echo "## (256, 256, $n) $p MFFT_C2R_MLDS=$m" | tee -a $O
      MFFT_C2R_MLDS=$m timeout 300 python scripts/meshprof.py 256 256 $n $p 2>&1 | grep "bwd_z" | tee -a $O
    done
  done
done
