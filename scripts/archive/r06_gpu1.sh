#!/bin/bash
# round 6, first GPU session: parity of the fused nonlinear operation, then the solver composed vs fused
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_nonlinear.py -x -q > gpurun_out/r06/nl_tests.log 2>&1
echo "nonlinear tests rc=$?" | tee -a gpurun_out/r06/nl_tests.log
tail -5 gpurun_out/r06/nl_tests.log
for M in 8 9; do
  for mode in "--composed" ""; do
    timeout 600 python examples/spectral_dns_device.py --M $M --steps 3 --stages $mode > gpurun_out/r06/dns_M${M}${mode}.log 2>&1
    echo "M=$M $mode rc=$?"; head -30 gpurun_out/r06/dns_M${M}${mode}.log
  done
done
