#!/bin/bash
# round 5, session 16: the strided kernels with their row offsets as base + (k TPT) stride where the rows lie one stride apart
# (fft_kernels.h rows_plain): kbench3 A/B in one process, the stage tests, a sweep against the round's own
timeout 600 tools/build/kbench3 walk 5 > gpurun_out/r05_kbench3_plain_rows.txt 2>&1; tail -70 gpurun_out/r05_kbench3_plain_rows.txt
timeout 900 python3 -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -x -q > gpurun_out/r05_t16a.log 2>&1; tail -3 gpurun_out/r05_t16a.log
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --out gpurun_out/r05_plain_sweep.txt > gpurun_out/r05_plain_sweep.log 2>&1
tail -45 gpurun_out/r05_plain_sweep.log
