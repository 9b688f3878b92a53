#!/bin/bash
# one rank: forward route in place (0) / y and x out of place (1) / z into the work buffer, y in place there, x out of place (2)
out=gpurun_out/r04_fwd_oop_ab.txt
: > $out
for rep in 1 2; do
for cfg in "1024 double" "512 double" "1024 single" "2048 single" "1536 double"; do
  set -- $cfg
  for m in 0 2 1; do
    echo -n "MFFT_FWD_OOP=$m " >> $out
    MFFT_FWD_OOP=$m python3 bench.py --size $1 --precision $2 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
  done
done
done
cat $out
