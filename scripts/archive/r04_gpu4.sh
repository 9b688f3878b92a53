#!/bin/bash
# round 4: wave-packed real kernels on the device (stage tests), their sizes in the sweep, then the A/B runs
python -m pytest tests/test_gpu_stages.py tests/test_gpu_cabi_from_c.py -x -q > gpurun_out/r04_stage_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_stage_tests.log
tail -4 gpurun_out/r04_stage_tests.log
out=gpurun_out/r04_wave_packed_sweep.txt
: > $out
for n in 288 500 576 600 720 900 1000 1152 1200 1440; do
  python3 bench.py --size $n --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
for n in 576 720 900 1000 1152; do
  python3 bench.py --size $n --precision single --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
done
cat $out
bash scripts/r04_gpu3.sh > /dev/null 2>&1
cat gpurun_out/r04_xpass_ab.txt
