#!/bin/bash
# round 5, session 25: the y-pass builds of 1440 / 1536 in double precision (registry.h register_col_ytile, core.hip launch_col)
# and LDS twiddles for 1536 in single: tests, the meshes concerned, the 3/2-rule pair of 1024^3 (1536-point passes), A/B by MFFT_YTILE
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t25a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t25a.log | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_line.py -x -q > gpurun_out/r05_t25b.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t25b.log | tail -2
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 1440 1536 --precisions fp64 fp32 --out gpurun_out/r05_ytile_sweep.txt > gpurun_out/r05_ytile_sweep.log 2>&1; tail -9 gpurun_out/r05_ytile_sweep.log
for rep in 1 2; do
  for y in 1 0; do echo "== MFFT_YTILE=$y (rep $rep)"; MFFT_YTILE=$y python3 scripts/padprof.py 1024 slab double; MFFT_YTILE=$y python3 scripts/padprof.py 960 slab double; done
done
python3 scripts/padprof.py 1024 slab single
python3 scripts/padprof.py 1024 X double
