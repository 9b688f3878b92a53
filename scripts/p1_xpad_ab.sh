#!/bin/bash
# A/B of the one-rank padded-plane route for real data (plan.hip p1_plane_pad; MFFT_P1_XPAD = cache lines, 0 = off).  Round 4.
out=gpurun_out/r04_p1_xpad_ab.txt
: > $out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_rank_padded_planes or mask_on_load or two_thirds_rule_pruned" 2>&1 | tail -3 >> $out
for rep in 1 2; do
for cfg in "256 double" "512 double" "512 single" "1024 single" "1024 double"; do
  set -- $cfg
  for m in 0 1 2 3; do
    echo -n "MFFT_P1_XPAD=$m " >> $out
    MFFT_P1_XPAD=$m python3 bench.py --size $1 --precision $2 --steps 20 --warmup 5 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
  done
done
done
cat $out
