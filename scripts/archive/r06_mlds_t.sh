#!/bin/bash
# c2r mirrors through LDS (C2RFft MLDS) on the 27 * 2^a row plans and the column-limited kernels of their 9 * 2^a neighbours:
# MFFT_C2R_MLDS = 0 never, 1 the shipped rule, 2 wherever built, 3 limited kernels only.  bwd_z of the plain and 3/2-rule pairs.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/mlds_t.txt
: > $O
for m in 0 2; do
  for n in 384 576 768 864 1152; do
    for p in double single; do
      echo "== MFFT_C2R_MLDS=$m" | tee -a $O
      MFFT_C2R_MLDS=$m timeout 300 python scripts/pitchprof.py $n $p none 2>&1 | grep -v "^$" | tee -a $O
    done
  done
  echo "== MFFT_C2R_MLDS=$m" | tee -a $O
  MFFT_C2R_MLDS=$m timeout 300 python scripts/pitchprof.py 1728 double none 2>&1 | grep -v "^$" | tee -a $O
  MFFT_C2R_MLDS=$m timeout 300 python scripts/pitchprof.py 1728 single none 2>&1 | grep -v "^$" | tee -a $O
done
