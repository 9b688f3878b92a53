#!/bin/bash
# "any n": lengths without a radix plan (one-workgroup chirp-z up to 4096, Bluestein over a four-step transform beyond) -- stage times and
# the fraction of the 8 TB/s roofline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/any_n_sweep.txt
: > $out
for m in "1009 1009 1009" "1120 1120 1120" "997 64 64" "64 997 64" "64 64 998" "5000 64 64" "64 5000 64" "64 64 5000" "10007 32 32" "32 10007 32" "32 32 10008" "10000 32 32" "2240 64 64" "64 64 2240"; do
  timeout 300 python scripts/meshprof.py $m double >> $out 2>&1
  timeout 300 python scripts/meshprof.py $m single >> $out 2>&1
done
cat $out
