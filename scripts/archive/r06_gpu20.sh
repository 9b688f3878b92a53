#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_stages.py -q -k "70 or 140 or 280 or 560 or 1120 or 2240" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_toolchain_canary.py -q 2>&1 | tail -2
for m in "1120 1120 1120" "560 560 560" "64 64 2240"; do timeout 300 python scripts/meshprof.py $m single 2>&1 | head -1; done
