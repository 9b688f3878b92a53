#!/bin/bash
out=gpurun_out/r04_r2c_rtw.txt
: > $out
for rep in 1 2; do
for cfg in "1024 double" "512 double" "1024 single" "2048 single" "768 double" "2048 double" "256 double"; do
  set -- $cfg
  python3 bench.py --size $1 --precision $2 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
done
cat $out
