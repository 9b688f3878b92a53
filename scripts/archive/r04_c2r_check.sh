#!/bin/bash
# after the c2r load changes (unconditional bins in the column-limited / z-chunked kernels, twiddles loaded with the bins): z stages
out=gpurun_out/r04_c2r_check.txt
: > $out
for cfg in "1024 double" "1536 double" "2048 double" "768 double" "1152 double" "1024 single" "1536 single" "2048 single"; do
  set -- $cfg
  python3 bench.py --size $1 --precision $2 --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
for prec in double single; do
  python3 scripts/padprof.py 1024 slab $prec >> $out 2>&1
  python3 scripts/padprof.py 1024 X $prec >> $out 2>&1
done
python3 scripts/padprof.py 512 slab double >> $out 2>&1
cat $out
