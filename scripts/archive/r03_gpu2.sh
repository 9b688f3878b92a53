#!/bin/bash
mkdir -p gpurun_out
run() {  # world size prio cus
  echo "== world=$1 size=$2 MFFT_COMM_PRIORITY=$3 MFFT_COMM_CUS=$4"
  MFFT_COMM_PRIORITY=$3 MFFT_COMM_CUS=$4 timeout 600 python bench.py --gpus $1 --size $2 --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off --transport ipc 2>gpurun_out/b_$1_$2_$3_$4.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['ms_per_step'], d['config']['exchange_pipeline_tuning_ms_per_pair'], d.get('degraded'))
"
}
run 8 128 0 0
run 8 128 1 16
run 4 128 1 0
run 4 128 0 0
run 2 128 1 0
run 2 128 0 0
run 8 1024 0 0
run 8 1024 1 0
run 4 1024 0 0
