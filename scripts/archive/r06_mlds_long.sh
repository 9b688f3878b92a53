#!/bin/bash
# c2r mirrors through LDS on rows of more than a wave's threads (real 3072 / 4096 / 6144 / 8192) and fp32 1728 capped for two workgroups
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/mlds_long.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "c2r or real or irfft or rfft" 2>&1 | tail -2 | tee -a $O
for n in 3072 4096 6144 8192; do
  for p in double single; do
    for m in 0 1; do
      echo "## n=$n $p MFFT_C2R_MLDS=$m" | tee -a $O
      MFFT_C2R_MLDS=$m timeout 300 python scripts/meshprof.py 256 256 $n $p 2>&1 | grep "bwd_z" | tee -a $O
    done
  done
done
for n in 2048 1536; do
  for p in double single; do
    for m in 0 1; do
      echo "## 3/2-rule (128, 128, $n) $p MFFT_C2R_MLDS=$m" | tee -a $O
      MFFT_C2R_MLDS=$m timeout 300 python scripts/meshprof.py 128 128 $n $p 3/2-rule 2>&1 | grep "bwd_z.*3/2" | tee -a $O
    done
  done
done
for p in single double; do timeout 300 python scripts/pitchprof.py 1728 $p none 2>&1 | grep plain | tee -a $O; done
for p in single double; do timeout 300 python scripts/pitchprof.py 1152 $p none 2>&1 | grep "3/2" | tee -a $O; done
timeout 300 python scripts/pitchprof.py 768 double none 2>&1 | tee -a $O
