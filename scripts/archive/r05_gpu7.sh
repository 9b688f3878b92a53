#!/bin/bash
# round 5, session 7: register caps for a second / third workgroup per CU (registry.h col_wgs, round-5 additions): the library
# before the change (reduced, under _ab/old) against the new one, alternating, one session
R=$PWD
out=$R/gpurun_out/r05_col_occupancy_caps.txt
: > $out
B="--steps 8 --warmup 3 --cpu-baseline off --pencil-extra off"
for rep in 1 2; do
for cfg in "1200 double" "1200 single" "720 single" "2304 single" "2400 single"; do
  set -- $cfg
  [ $rep = 2 ] && [ $1 -ge 2304 ] && continue
  for lib in old new; do
    if [ $lib = old ]; then cd $R/_ab/old; else cd $R; fi
    echo "== $lib $1 $2 (rep $rep)" >> $out
    python3 bench.py --size $1 --precision $2 $B 2>/dev/null | python3 $R/scripts/show_bench.py >> $out
  done
done
done
cd $R
cat $out
python3 -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05_gputests_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05_gputests_final.log
tail -12 gpurun_out/r05_gputests_final.log
