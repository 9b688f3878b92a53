#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite, then the new 7 * 2^a sizes and the headline line
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gputests.log
tail -5 gpurun_out/r04_gputests.log
out=gpurun_out/r04_radix7_sweep.txt
: > $out
for n in 448 896 1792; do
  python3 bench.py --size $n --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
for n in 896 1792; do
  python3 bench.py --size $n --precision single --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
cat $out
python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py | tee gpurun_out/r04_headline.txt
