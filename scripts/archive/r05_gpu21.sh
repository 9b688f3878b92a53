#!/bin/bash
# round 5, session 21: 2048 fp32 without its non-temporal variants (they overran the 64-register cap); 1440 / 1200 fp32 meshes
timeout 600 python3 -m pytest tests/test_gpu_stages.py -x -q -k "2048 or 1440 or 4096" > gpurun_out/r05_t21a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t21a.log | tail -2
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 2048 1440 1200 --precisions fp32 --out gpurun_out/r05_narrow_sweep4.txt > gpurun_out/r05_narrow_sweep4.log 2>&1; tail -9 gpurun_out/r05_narrow_sweep4.log
