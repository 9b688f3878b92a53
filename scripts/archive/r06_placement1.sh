#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r06
out=gpurun_out/r06/placement_modes.txt
: > $out
for n in 800 768; do for i in 1 2 3 4 5 6 7 8 9 10; do timeout 120 python scripts/placement_modes.py $n >> $out 2>&1; done; done
cat $out
timeout 120 rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TCC\|TCA\|TA\|TD\|GRBM\|SQ\|GL2C\|UTCL2\|ATC\|MC\|VM\)_[A-Z0-9_]*" | sort -u > gpurun_out/r06/counter_names.txt
wc -l gpurun_out/r06/counter_names.txt; grep -i "utcl\|tlb\|xnack\|TCP_.*MISS\|TCC_.*STALL\|LATENCY" gpurun_out/r06/counter_names.txt | head -80
