#!/bin/bash
python3 scripts/xpass_kernel_ab.py 2>&1 | tee gpurun_out/r04_xpass_kernel_ab.txt
python -m pytest tests -m gpu -q > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gputests.log
tail -6 gpurun_out/r04_gputests.log
