#!/bin/bash
# round 5, session 27: 256-thread workgroups with the split exchange for the contiguous-axis kernels of the 42-values plans
# in double precision (registry.h row_wide42): stage tests, parity + fuzz + line, the 21 * 2^a meshes
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t27a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t27a.log | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_line.py -x -q > gpurun_out/r05_t27b.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t27b.log | tail -2
python3 scripts/perf_gate.py --baseline gpurun_out/r05_small_sweep42.txt --out gpurun_out/r05_wide42_sweep.txt > gpurun_out/r05_wide42_sweep.log 2>&1; tail -16 gpurun_out/r05_wide42_sweep.log
