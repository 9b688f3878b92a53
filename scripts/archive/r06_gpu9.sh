#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=gpurun_out/r06/pitched_ab.txt
: > $out
for rep in 1 2; do
timeout 300 python scripts/pitchprof.py 1024 double none auto >> $out 2>&1
done
timeout 300 python scripts/pitchprof.py 512 double none auto >> $out 2>&1
timeout 300 python scripts/pitchprof.py 768 double none auto >> $out 2>&1
timeout 300 python scripts/pitchprof.py 1024 single none auto >> $out 2>&1
for m in "" "--pitched"; do timeout 300 python examples/spectral_dns_device.py --M 9 --steps 3 --stages $m >> $out 2>&1; done
cat $out
