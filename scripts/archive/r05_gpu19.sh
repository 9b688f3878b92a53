#!/bin/bash
# round 5, session 19: single precision 1792 on 64-byte tiles with two workgroups per CU (registry.h col_narrow_f32); the
# 3/2-rule image of 800^3 and the 2/3-rule at 1200^3 with 16 and with 8 columns per workgroup (alt library: the one object with
# -DMFFT_COL_WIDE=0, tools/build/alt)
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t19a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t19a.log | tail -2
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 896 1792 --precisions fp32 --out gpurun_out/r05_narrow_sweep.txt > gpurun_out/r05_narrow_sweep.log 2>&1; tail -8 gpurun_out/r05_narrow_sweep.log
for rep in 1 2; do
  echo "== 16 columns (rep $rep)"; python3 scripts/padprof.py 800 slab double; python3 scripts/dealias23_time.py 1200 double
  cp mpifft4py_amd/libmpifft4py_amd.so /tmp/lib_main.so; cp tools/build/alt/libmpifft4py_amd.so mpifft4py_amd/libmpifft4py_amd.so
  echo "== 8 columns (rep $rep)"; python3 scripts/padprof.py 800 slab double; python3 scripts/dealias23_time.py 1200 double
  cp /tmp/lib_main.so mpifft4py_amd/libmpifft4py_amd.so
done
