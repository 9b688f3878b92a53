#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/nlz3_prefetch.txt
: > $out
for v in 0 1 2 0 1 2; do
  MFFT_NLZ3=$v timeout 300 python scripts/nlz_bench.py 768 257 73728 double 1536 513 36864 double 384 129 147456 double 768 257 73728 single 1536 513 73728 single >> $out 2>&1
done
cat $out
MFFT_NLZ3=2 timeout 600 python -m pytest tests/test_gpu_nonlinear.py -x -q 2>&1 | tail -2
for v in 0 2; do MFFT_NLZ3=$v timeout 300 python examples/spectral_dns_device.py --M 9 --steps 3 --stages 2>&1 | grep -v "fwd_[xyz] "; done
