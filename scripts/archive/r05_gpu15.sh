#!/bin/bash
# round 5, session 15: the miscompile cure moved to where the fault is (fft_core.h MFFT_LAUNDER_MODE 4): every length through the
# stage tests, the parity and fuzz suites, then the meshes whose contiguous-axis kernels carry it, against the round's own sweep
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t15a.log 2>&1; tail -3 gpurun_out/r05_t15a.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_line.py tests/test_gpu_demo.py -x -q > gpurun_out/r05_t15b.log 2>&1; tail -3 gpurun_out/r05_t15b.log
python3 scripts/perf_gate.py --baseline profiles/r05_size_sweep.txt --sizes 480 600 720 900 960 1200 1440 --out gpurun_out/r05_cure_sweep.txt > gpurun_out/r05_cure_sweep.log 2>&1
tail -16 gpurun_out/r05_cure_sweep.log
python3 scripts/perf_gate.py --baseline profiles/r05_radix42_sweep.txt --out gpurun_out/r05_cure_sweep42.txt > gpurun_out/r05_cure_sweep42.log 2>&1
tail -12 gpurun_out/r05_cure_sweep42.log
