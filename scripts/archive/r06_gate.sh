#!/bin/bash
# the perf gate twice in a row on an unchanged tree (green twice), then the canary test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1700 python3 scripts/perf_gate.py --out gpurun_out/r06/size_sweep_1.txt > gpurun_out/r06/perf_gate_1.log 2>&1; echo "gate 1 rc=$?"
tail -50 gpurun_out/r06/perf_gate_1.log
timeout 1700 python3 scripts/perf_gate.py --out gpurun_out/r06/size_sweep_2.txt > gpurun_out/r06/perf_gate_2.log 2>&1; echo "gate 2 rc=$?"
tail -8 gpurun_out/r06/perf_gate_2.log
timeout 600 python -m pytest tests/test_gpu_toolchain_canary.py -q 2>&1 | tail -5
