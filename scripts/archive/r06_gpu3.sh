#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/nlz3_variants.txt
: > $out
for v in 0 11 12 13 14 15 16 21 22; do
  MFFT_NLZ_VARIANT=$v timeout 300 python scripts/nlz_bench.py 768 257 73728 double 1536 513 36864 double 512 257 65536 double 1024 513 49152 double >> $out 2>&1
done
MFFT_NLZ3=0 timeout 300 python scripts/nlz_bench.py 768 257 73728 double 1536 513 36864 double >> $out 2>&1
timeout 300 python scripts/nlz_bench.py 768 257 73728 single 1536 513 73728 single 1024 513 98304 single >> $out 2>&1
MFFT_NLZ3=0 timeout 300 python scripts/nlz_bench.py 768 257 73728 single 1536 513 73728 single >> $out 2>&1
cat $out
timeout 900 python -m pytest tests/test_gpu_nonlinear.py -x -q 2>&1 | tail -3
