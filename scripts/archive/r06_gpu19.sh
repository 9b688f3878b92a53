#!/bin/bash
# 35 * 2^a in single precision: radix-70 plans against the chirp-z route (MFFT_NO_PLANS_S is not a switch: compare with profiles/r06_any_n_sweep.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "70 or 140 or 280 or 560 or 1120 or 2240" 2>&1 | tail -4
out=gpurun_out/r06/radix70.txt
: > $out
for m in "1120 1120 1120" "560 560 560" "2240 64 64" "64 2240 64" "64 64 2240" "280 280 280"; do
  timeout 300 python scripts/meshprof.py $m single >> $out 2>&1
done
cat $out
timeout 300 python scripts/padprof.py 1120 slab single 2>&1 | tail -4
