#!/bin/bash
# round 5, session 6: the whole suite again (after the obsolete unsupported-length checks), fuzz with new seeds
python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05_gputests_final.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05_gputests_final.log
tail -14 gpurun_out/r05_gputests_final.log
for seed in 501 502 503; do timeout 400 python3 scripts/fuzz_parity.py 40 $seed 2>&1 | tail -2; done > gpurun_out/r05_fuzz.log 2>&1
for seed in 601 602; do timeout 300 python3 scripts/fuzz_stages.py 150 $seed 2>&1 | tail -2; done >> gpurun_out/r05_fuzz.log 2>&1
cat gpurun_out/r05_fuzz.log
