#!/bin/bash
# Before / after of plans.h group T (27 * 2^a) and the wider c2r-through-LDS rule: _ab/prev = the library of the commit before
# (git archive + make), the tree = this build.  Plain pairs of 432 / 864 / 1728, 3/2-rule pairs of 288 / 576 / 768 / 1152, Taylor-Green at 576^3.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/group_t_ab.txt
: > $O
run() {      # label, tree
  for n in 432 864; do for p in double single; do
    echo "## $1" | tee -a $O; timeout 600 python $2/scripts/pitchprof.py $n $p none 2>&1 | grep plain | tee -a $O; done; done
  for p in double single; do echo "## $1" | tee -a $O; timeout 900 python $2/scripts/pitchprof.py 1728 $p none 2>&1 | grep plain | tee -a $O; done
  for n in 288 576 768; do for p in double single; do
    echo "## $1" | tee -a $O; timeout 600 python $2/scripts/pitchprof.py $n $p none 2>&1 | grep "3/2" | tee -a $O; done; done
  echo "## $1" | tee -a $O; timeout 900 python $2/scripts/pitchprof.py 1152 single none 2>&1 | grep "3/2" | tee -a $O
  echo "## $1" | tee -a $O; timeout 900 python $2/examples/spectral_dns_device.py --N 576 --steps 3 2>&1 | grep "RK4" | tee -a $O
  echo "## $1" | tee -a $O; timeout 900 python $2/examples/spectral_dns_device.py --N 576 --steps 3 --dealias None 2>&1 | grep "RK4" | tee -a $O
}
run before _ab/prev
run after .
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_nonlinear.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded" 2>&1 | tail -2 | tee -a $O
