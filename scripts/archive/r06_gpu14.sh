#!/bin/bash
# c2r with the mirrors through LDS (C2RFft MLDS) against the second load, z stage of (256, 256, n) meshes and of two cubes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/c2r_mlds.txt
: > $out
for n in 900 1000 1152 1440 1800 2000 2304 2880 3600 4608; do
  for prec in double single; do
    for m in 0 1; do
      echo "## n=$n $prec MFFT_C2R_MLDS=$m" >> $out
      MFFT_C2R_MLDS=$m timeout 300 python scripts/meshprof.py 256 256 $n $prec 2>&1 | grep "mesh\|bwd_z" >> $out
    done
  done
done
for n in 1440 900; do for m in 0 1 0 1; do echo "## cube $n MFFT_C2R_MLDS=$m" >> $out; MFFT_C2R_MLDS=$m timeout 300 python scripts/meshprof.py $n $n $n double 2>&1 | grep "mesh\|bwd_z" >> $out; done; done
grep -A2 "^##" $out | grep "##\|bwd_z" | paste - - | awk '{print $2, $3, $4, $6, $7}'
MFFT_C2R_MLDS=1 timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "c2r or real or irfft" 2>&1 | tail -2
