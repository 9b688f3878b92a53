#!/bin/bash
# round 5, session 11: non-temporal builds of the pad-on-load / truncate-on-store kernels on the line-aligned y passes
out=gpurun_out/r05_pad_nt_ab.txt
: > $out
for rep in 1 2 3; do
  for nt in 1 0; do
    echo "== MFFT_NT=$nt (rep $rep)" >> $out
    MFFT_NT=$nt python3 scripts/padprof.py 1024 slab double >> $out 2>&1
  done
done
for nt in 1 0; do
  echo "== 1536 (padded 2304) MFFT_NT=$nt" >> $out
  MFFT_NT=$nt python3 scripts/padprof.py 1536 slab double >> $out 2>&1
done
cat $out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "padded or three_halves" > gpurun_out/r05_t11.log 2>&1; tail -3 gpurun_out/r05_t11.log
