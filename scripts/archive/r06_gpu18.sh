#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_multiprocess.py -x -q -k "process_per_rank_parity" 2>&1 | tail -8
