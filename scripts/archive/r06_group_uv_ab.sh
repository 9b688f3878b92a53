#!/bin/bash
# Before / after of plans.h groups U (135 * 2^a, 1350 / 2700 / 2250, 675 / 1125) and V (81 * 2^a): _ab/prev = the library of the commit
# before, the tree = this build.  3/2-rule pairs of 360 / 720 / 1440 / 900 / 432 / 864, plain pairs of 1080 / 648 / 1296.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/group_uv_ab.txt
: > $O
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_nonlinear.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded" 2>&1 | tail -2 | tee -a $O
run() {      # label, tree
  for n in 360 720 900 432 864; do
    echo "## $1" | tee -a $O; timeout 600 python $2/scripts/pitchprof.py $n double none 2>&1 | grep "3/2" | tee -a $O; done
  for n in 720 864; do
    echo "## $1" | tee -a $O; timeout 600 python $2/scripts/pitchprof.py $n single none 2>&1 | grep "3/2" | tee -a $O; done
  echo "## $1" | tee -a $O; timeout 900 python $2/scripts/pitchprof.py 1440 single none 2>&1 | grep "3/2" | tee -a $O
  for n in 648 1080 1296; do
    echo "## $1" | tee -a $O; timeout 600 python $2/scripts/pitchprof.py $n double none 2>&1 | grep "plain" | tee -a $O; done
  echo "## $1" | tee -a $O; timeout 900 python $2/examples/spectral_dns_device.py --N 432 --steps 3 2>&1 | grep "RK4" | tee -a $O
}
[ -f _ab/prev/mpifft4py_amd/libmpifft4py_amd.so ] && run before _ab/prev
run after .
