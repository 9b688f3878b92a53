#!/bin/bash
# Taylor-Green loop under the 2/3-rule: fused nonlinear operation (mask on load, every kz carried) against the composition (pruned
# inverse passes), pitched and compact spectra
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/dns23.txt
: > $O
for M in 8 9; do
  for args in "" "--compact" "--composed"; do
    echo "## --M $M --dealias 2/3-rule $args" | tee -a $O
    timeout 600 python examples/spectral_dns_device.py --M $M --steps 3 --dealias 2/3-rule --stages $args 2>&1 | grep -v "0 launches" | tee -a $O
  done
done
