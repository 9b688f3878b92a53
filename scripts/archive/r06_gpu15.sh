#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_nonlinear.py tests/test_gpu_demo.py tests/test_gpu_pitched.py -x -q -k "not 512" > gpurun_out/r06/t15.log 2>&1; echo rc=$?; tail -12 gpurun_out/r06/t15.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "assigned_on_one_rank or edited_in_place" 2>&1 | tail -3
for m in 8 9; do timeout 300 python examples/spectral_dns_device.py --M $m --steps 3 --stages 2>&1 | grep -v "fwd_[xyz] "; done
timeout 300 python examples/spectral_dns_device.py --M 8 --steps 3 --stages --ranks 2 2>&1 | grep -v "fwd_[xyz] \|bwd_"
timeout 300 python examples/spectral_dns_device.py --M 8 --steps 3 --stages --ranks 2 --composed 2>&1 | tail -3
