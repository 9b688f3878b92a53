#!/bin/bash
# the communication stream of pipelined plans ran at the LOWEST priority while the notes said "normal".
# bench.py --gpus P --size N --transport ipc, all ranks on ONE GPU, ms per pair by exchange pipeline depth, with the
# communication stream at the lowest (default, unset), the normal (0) and the highest (1) stream priority.
mkdir -p gpurun_out
out=gpurun_out/r04_comm_priority.txt
: > $out
run() {  # world size prio
  echo "== world=$1 size=$2 MFFT_COMM_PRIORITY=${3:-unset (lowest)}" >> $out
  if [ -n "$3" ]; then export MFFT_COMM_PRIORITY=$3; else unset MFFT_COMM_PRIORITY; fi
  timeout 600 python bench.py --gpus $1 --size $2 --steps 3 --warmup 1 --cpu-baseline off --pencil-extra off --transport ipc 2>gpurun_out/bp_$1_$2_$3.err | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    t = d['config']['exchange_pipeline_tuning_ms_per_pair']
    print(round(d['ms_per_step'], 2), {k: {kk: round(vv, 2) for kk, vv in v.items()} if isinstance(v, dict) else v for k, v in t.items() if k == 'ipc'}, d.get('degraded'))
" >> $out
}
for p in "" 0 1; do run 2 128 $p; done
for p in "" 0 1; do run 4 128 $p; done
for p in "" 0 1; do run 2 512 $p; done
for p in "" 0 1; do run 4 1024 $p; done
cat $out
