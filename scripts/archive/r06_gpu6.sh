#!/bin/bash
# round 6: full GPU suite, bench line, the solver's kernel trace at 512^3, a 1024^3 3/2-rule step on one GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06/gputests.log 2>&1
tail -4 gpurun_out/r06/gputests.log
timeout 900 python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err; echo "bench rc=$?"
python scripts/show_bench.py gpurun_out/r06/bench.json 2>/dev/null | head -40 || head -c 3000 gpurun_out/r06/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/dnsprof -- python3 $R/examples/spectral_dns_device.py --M 9 --steps 3 > $R/gpurun_out/r06/dnsprof.log 2>&1
cd $R
f=$(find gpurun_out/r06/dnsprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r06/dns_512_kernel_stats.csv && head -20 $f | cut -c1-220
find gpurun_out/r06/dnsprof -name "*kernel_trace.csv" -delete
timeout 900 python examples/spectral_dns_device.py --M 10 --steps 1 --stages > gpurun_out/r06/dns_1024.log 2>&1; echo "1024 rc=$?"; cat gpurun_out/r06/dns_1024.log | tail -12
