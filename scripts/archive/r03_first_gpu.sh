#!/bin/bash
# round 3, first GPU call: CU-mask / overlap probe, the multi-process tests with the new pull modes, and what the
# number of hardware queues does to 8 processes on one GPU
mkdir -p gpurun_out
tools/build/overlap_probe 8 16 32 > gpurun_out/cu_mask_probe.txt 2>&1
echo "probe rc $?"
tail -12 gpurun_out/cu_mask_probe.txt
timeout 1500 python -m pytest tests/test_gpu_multiprocess.py -x -q -k "parity" 2>&1 | tail -15
for q in default 2 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for mode in kernel copy streams; do
    echo "== GPU_MAX_HW_QUEUES=$q MFFT_IPC_PULL=$mode"
    MFFT_IPC_PULL=$mode timeout 300 python bench.py --gpus 8 --size 128 --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off --transport ipc 2>gpurun_out/bench8_${q}_${mode}.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['ms_per_step'], d['config']['exchange_pipeline_tuning_ms_per_pair'], d.get('degraded'))
"
  done
done
