#!/bin/bash
# A/B of the one-rank real transform's line-aligned intermediate (plan.hip aligned_route; developer tool, round 4)
out=gpurun_out/r04_aligned_ab.txt
: > $out
for rep in 1 2; do
for n in 1024 512; do
for m in 0 1 2 3; do
  echo -n "MFFT_ALIGNED=$m " >> $out
  MFFT_ALIGNED=$m python3 bench.py --size $n --steps 20 --warmup 5 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
done
done
MFFT_ALIGNED=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab_r2c or golden" 2>&1 | tail -3 >> $out
cat $out
