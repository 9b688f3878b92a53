#!/bin/bash
# the solver's kernel trace at 512^3 and a 1024^3 3/2-rule step on one GPU (every command under its own timeout)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r06
timeout 600 python examples/spectral_dns_device.py --M 10 --steps 1 --stages > gpurun_out/r06/dns_1024.log 2>&1; echo "1024 rc=$?"; tail -12 gpurun_out/r06/dns_1024.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/dnsprof -- python3 $R/examples/spectral_dns_device.py --M 9 --steps 3 > $R/gpurun_out/r06/dnsprof.log 2>&1; echo "prof rc=$?"
cd $R
tail -3 gpurun_out/r06/dnsprof.log
f=$(find gpurun_out/r06/dnsprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r06/dns_512_kernel_stats.csv && head -20 $f | cut -c1-200
find gpurun_out/r06/dnsprof -name "*kernel_trace.csv" -delete
