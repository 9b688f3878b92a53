#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 2 4; do
timeout 300 python examples/spectral_dns_device.py --M 8 --steps 3 --stages --ranks $r 2>&1 | grep -v " 0.000 ms"
timeout 300 python examples/spectral_dns_device.py --M 8 --steps 3 --stages --ranks $r --composed 2>&1 | grep "^N =\|^k"
done
