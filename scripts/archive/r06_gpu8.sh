#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_pitched.py tests/test_gpu_nonlinear.py -x -q > gpurun_out/r06/pitched_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r06/pitched_tests.log
