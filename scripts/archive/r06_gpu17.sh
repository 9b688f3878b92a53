#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_pitched.py tests/test_gpu_demo.py -x -q 2>&1 | tail -3
timeout 900 python examples/spectral_dns_device.py --M 10 --steps 2 --stages > gpurun_out/r06/dns_1024_final.log 2>&1; echo "1024 rc=$?"; grep -v " 0.000 ms" gpurun_out/r06/dns_1024_final.log
timeout 300 python examples/spectral_dns_device.py --M 9 --steps 5 --stages 2>&1 | grep -v " 0.000 ms"
timeout 300 python examples/spectral_dns_device.py --M 9 --steps 5 --stages --composed 2>&1 | grep "^N =\|^k"
