#!/bin/bash
# round 5, session 1: (a) bisect of the radix-7 c2r regression (reduced libraries of five commits under _bisect/),
# (b) r03 library against HEAD at 576 / 768 / 800 / 1152 in one session, (c) test durations
R=$PWD
out=$R/gpurun_out/r05_bisect.txt
: > $out
B="--steps 10 --warmup 3 --cpu-baseline off --pencil-extra off"
for rep in 1 2; do
for c in cdf2b38 ac37db9 af5c707 e284226 HEAD; do
  cd $R/_bisect/$c
  echo "== $c 896 (rep $rep)" >> $out
  python3 bench.py --size 896 $B 2>>$out.err | python3 $R/scripts/show_bench.py >> $out
done
done
for c in cdf2b38 HEAD; do
  cd $R/_bisect/$c
  echo "== $c 1792" >> $out
  python3 bench.py --size 1792 $B 2>>$out.err | python3 $R/scripts/show_bench.py >> $out
done
cd $R
echo "== main full library 896" >> $out
python3 bench.py --size 896 $B 2>>$out.err | python3 scripts/show_bench.py >> $out
out2=$R/gpurun_out/r05_r03_vs_head.txt
: > $out2
for rep in 1 2; do
for n in 576 768 800 1152; do
  cd $R/_bisect/4dcb4d8
  echo "== r03 $n" >> $out2
  python3 bench.py --size $n $B 2>>$out2.err | python3 $R/scripts/show_bench.py >> $out2
  cd $R
  echo "== HEAD $n" >> $out2
  python3 bench.py --size $n $B 2>>$out2.err | python3 scripts/show_bench.py >> $out2
done
done
cd $R
cat $out $out2
timeout 1100 python3 -m pytest tests -m gpu -x -q --durations=150 > gpurun_out/r05_gputests_durations.log 2>&1
tail -5 gpurun_out/r05_gputests_durations.log
