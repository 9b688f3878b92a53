#!/bin/bash
# round 5, session 26: 256-thread workgroups on 256- / 512-byte tiles for the plans with 8 or 16 threads per transform
# (registry.h col_wide_small): all stage tests, parity + fuzz, the meshes concerned
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t26a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t26a.log | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_line.py tests/test_gpu_demo.py -x -q > gpurun_out/r05_t26b.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t26b.log | tail -2
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 448 480 --precisions fp64 --out gpurun_out/r05_small_sweep.txt > gpurun_out/r05_small_sweep.log 2>&1; tail -7 gpurun_out/r05_small_sweep.log
python3 scripts/perf_gate.py --baseline profiles/r05_radix42_sweep.txt --out gpurun_out/r05_small_sweep42.txt > gpurun_out/r05_small_sweep42.log 2>&1; tail -11 gpurun_out/r05_small_sweep42.log
for n in 240 320; do python3 bench.py --size $n --steps 50 --warmup 10 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py; done
