#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_multiprocess.py -x -q 2>&1 | tail -15
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for fl in 0 1; do
    echo "== GPU_MAX_HW_QUEUES=$q MFFT_IPC_STREAM_FLAGS=$fl (streams pull mode only matters)"
    MFFT_IPC_STREAM_FLAGS=$fl timeout 900 python bench.py --gpus 8 --size 128 --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off --transport ipc 2>gpurun_out/b8s_${q}_${fl}.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    t = d['config']['exchange_pipeline_tuning_ms_per_pair']
    print(d['ms_per_step'], d['config']['exchange_transport'], d['config']['exchange_pipeline_depth'], d.get('degraded'))
    for k, v in t.items(): print('  ', k, v)
"
  done
done
