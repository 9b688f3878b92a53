#!/bin/bash
# round 5, session 22: the new parity cases (tile shapes over ranks), the bare tile pattern of 1024 on wider tiles, LDS twiddles
# for the single-precision kernels that run several workgroups per CU without them
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "wide_and_narrow or padded_long_axes" > gpurun_out/r05_t22a.log 2>&1; grep -n "passed\|failed" gpurun_out/r05_t22a.log | tail -2
timeout 200 tools/build/membench tile1024w > gpurun_out/r05_membench_tile1024w.txt 2>&1; cat gpurun_out/r05_membench_tile1024w.txt
timeout 600 tools/build/kbench3 tw1536 3 > gpurun_out/r05_kbench3_tw1536.txt 2>&1; grep -v "check" gpurun_out/r05_kbench3_tw1536.txt; grep -c MISMATCH gpurun_out/r05_kbench3_tw1536.txt
