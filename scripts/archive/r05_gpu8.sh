#!/bin/bash
# round 5, session 8: register caps of the contiguous-axis kernels (registry.h row_occ_r5): the library before them (_ab/old)
# against the new one, alternating; then the whole suite
R=$PWD
out=$R/gpurun_out/r05_row_occupancy_caps.txt
: > $out
B="--steps 8 --warmup 3 --cpu-baseline off --pencil-extra off"
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then cd $R/_ab/old; else cd $R; fi
    echo "== $lib (rep $rep)" >> $out
    python3 bench.py --size 2048 --precision single $B 2>/dev/null | python3 scripts/show_bench.py >> $out
    python3 bench.py --size 1000 --precision double $B 2>/dev/null | python3 scripts/show_bench.py >> $out
    for cfg in "720 single" "1200 single" "800 double" "500 double"; do python3 scripts/c2cprof.py $cfg >> $out 2>&1; done
  done
done
cd $R
cat $out
python3 -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05_gputests_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05_gputests_final.log
tail -12 gpurun_out/r05_gputests_final.log
