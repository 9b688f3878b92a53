#!/bin/bash
# round 5, session 18: 16 columns per workgroup for the strided kernel of 1200 in double precision (registry.h col_wide):
# tests of the length, the meshes that use it (1200^3, the 3/2-rule image of 800^3, the 2/3-rule at 1200^3), kbench3 variants
timeout 600 python3 -m pytest tests/test_gpu_stages.py -x -q -k "1200 or 2400 or 800" > gpurun_out/r05_t18a.log 2>&1; tail -2 gpurun_out/r05_t18a.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r05_t18b.log 2>&1; tail -2 gpurun_out/r05_t18b.log
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 1200 --precisions fp64 fp32 --out gpurun_out/r05_wide_sweep.txt > gpurun_out/r05_wide_sweep.log 2>&1; tail -8 gpurun_out/r05_wide_sweep.log
python3 scripts/padprof.py 800 slab double
python3 scripts/dealias23_time.py 1200 double
timeout 500 tools/build/kbench3 occ1200 3 > gpurun_out/r05_kbench3_occ1200b.txt 2>&1; grep -v "check\|occupancy" gpurun_out/r05_kbench3_occ1200b.txt | tail -40; grep -c MISMATCH gpurun_out/r05_kbench3_occ1200b.txt
