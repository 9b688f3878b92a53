#!/bin/bash
# round 5, session 2: radix-7 cap verified through the perf gate; allocation-shift probe at 800^3; new dealias tests; suite time
R=$PWD
python3 scripts/perf_gate.py --baseline profiles/r04_radix7_sweep.txt --out gpurun_out/r05_radix7_gate.txt > gpurun_out/r05_radix7_gate.log 2>&1
echo "gate rc=$?" >> gpurun_out/r05_radix7_gate.log
cat gpurun_out/r05_radix7_gate.log
out=gpurun_out/r05_alloc_shift_probe.txt
: > $out
B="--steps 10 --warmup 3 --cpu-baseline off --pencil-extra off"
for n in 800 768; do
for mb in 0 1 64 300 1000 0 64; do
  echo "== $n^3 shift $mb MiB" >> $out
  MFFT_BENCH_ALLOC_SHIFT_MB=$mb python3 bench.py --size $n $B 2>/dev/null | python3 scripts/show_bench.py >> $out
done
done
cat $out
timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "one_element or edited_in_place or large_filter" > gpurun_out/r05_dealias_tests.log 2>&1
tail -5 gpurun_out/r05_dealias_tests.log
timeout 1100 python3 -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r05_gputests2.log 2>&1
tail -32 gpurun_out/r05_gputests2.log
