#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/nlz_wave.txt
: > $out
for v in 0 30 0 30; do
  MFFT_NLZ_VARIANT=$v timeout 300 python scripts/nlz_bench.py 768 257 73728 double 512 257 65536 double 384 129 147456 double 256 129 131072 double 768 257 73728 single 512 257 131072 single >> $out 2>&1
done
cat $out
timeout 600 python -m pytest tests/test_gpu_nonlinear.py -x -q 2>&1 | tail -2
for m in 8 9; do timeout 300 python examples/spectral_dns_device.py --M $m --steps 3 --stages 2>&1 | grep -v "fwd_[xyz] "; done
timeout 300 python scripts/meshprof.py 1009 1009 1010 double >> gpurun_out/r06/any_n_sweep.txt 2>&1
timeout 300 python scripts/meshprof.py 1009 1009 1010 single >> gpurun_out/r06/any_n_sweep.txt 2>&1
grep "mesh.*1009" gpurun_out/r06/any_n_sweep.txt
