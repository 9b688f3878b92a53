#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/nlz_pd.txt
: > $out
for v in 0 41 42 43 0 41; do
  MFFT_NLZ_VARIANT=$v timeout 300 python scripts/nlz_bench.py 768 257 73728 double 1536 513 36864 double 512 257 65536 double 1024 513 49152 double >> $out 2>&1
done
cat $out
