#!/bin/bash
# round 5, session 20: single precision 1440 / 1792 / 2048 on 64-byte tiles with two workgroups per CU (registry.h
# col_narrow_f32): all stage tests, parity + fuzz, the meshes concerned against the round's sweep
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t20a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t20a.log | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_line.py -x -q > gpurun_out/r05_t20b.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t20b.log | tail -2
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 1024 1536 2048 1792 --precisions fp32 --out gpurun_out/r05_narrow_sweep2.txt > gpurun_out/r05_narrow_sweep2.log 2>&1; tail -10 gpurun_out/r05_narrow_sweep2.log
python3 scripts/perf_gate.py --baseline gpurun_out/r05_cure_sweep42.txt --sizes 1440 1200 --precisions fp32 --out gpurun_out/r05_narrow_sweep3.txt > gpurun_out/r05_narrow_sweep3.log 2>&1; tail -8 gpurun_out/r05_narrow_sweep3.log
for prec in single; do python3 scripts/padprof.py 1024 slab $prec; done
