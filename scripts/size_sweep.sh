#!/bin/bash
# bench.py over a range of cube sizes and both precisions (one line per size; developer tool)
out=gpurun_out/size_sweep.txt
: > $out
for n in 64 128 256 384 448 500 512 576 640 768 800 896 1000 1024 1152 1280 1536 1792 480 600 720 900 960 1200 1440; do
  python3 bench.py --size $n --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
for n in 512 720 896 1024 1536 2048; do
  python3 bench.py --size $n --precision single --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
python3 bench.py --size 2048 --steps 5 --warmup 2 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
cat $out
