#!/bin/bash
# 27 * 2^a plans (plans.h group T): tests of the new lengths, then the pairs that use them -- plain 432 / 864 / 1728, the 3/2-rule
# pairs of 288 / 576 / 1152 (which had no fused pad / truncate passes before) and the Taylor-Green loop at 576^3.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/group_t.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_stages.py -q -x -m gpu -k "54 or 108 or 216 or 432 or 864 or 1728 or 3456" 2>&1 | tail -3 | tee -a $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "test_padded_long_axes" 2>&1 | tail -3 | tee -a $O
timeout 900 python -m pytest tests/test_gpu_nonlinear.py -q -x -m gpu -k "test_nonlinear_cross_one_rank" 2>&1 | tail -3 | tee -a $O
for n in 432 864 1728; do
  for p in double single; do timeout 300 python scripts/pitchprof.py $n $p none 2>&1 | grep -v "^$" | tee -a $O; done
done
for n in 288 576 1152; do
  for p in double single; do timeout 300 python scripts/pitchprof.py $n $p none auto 2>&1 | grep "3/2" | tee -a $O; done
done
timeout 600 python examples/spectral_dns_device.py --N 576 --steps 3 --stages 2>&1 | tee -a $O
timeout 600 python examples/spectral_dns_device.py --N 576 --steps 3 --composed 2>&1 | tail -2 | tee -a $O
timeout 600 python examples/spectral_dns_device.py --N 576 --steps 3 --dealias None --stages 2>&1 | tee -a $O
timeout 600 python examples/spectral_dns_device.py --N 576 --steps 3 --dealias None --composed 2>&1 | tail -2 | tee -a $O
