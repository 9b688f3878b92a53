#!/bin/bash
# round 5, session 13: radix plans of 21 * 2^a (plans.h group R, 42 values per thread): parity, then 672^3 / 1344^3 / 336^3 in both
# precisions (before: the one-workgroup chirp-z kernels; profiles/r04 notes: 672^3 fp32 13.1 - 13.6 ms, fp64 16.8 - 17.0, 1344^3 fp32 123.6)
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q -k "every_length or arbitrary_length" > gpurun_out/r05_t13.log 2>&1; tail -3 gpurun_out/r05_t13.log
out=gpurun_out/r05_radix42_sweep.txt
: > $out
B="--steps 8 --warmup 3 --cpu-baseline off --pencil-extra off"
for cfg in "336 double" "672 double" "1344 double" "336 single" "672 single" "1344 single" "2688 single"; do
  set -- $cfg
  python3 bench.py --size $1 --precision $2 $B 2>/dev/null | python3 scripts/show_bench.py >> $out
done
cat $out
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r05_t13b.log 2>&1; tail -3 gpurun_out/r05_t13b.log
