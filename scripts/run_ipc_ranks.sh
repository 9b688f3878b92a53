#!/bin/bash
export MFFT_TRANSPORT=ipc MFFT_LOCAL_TIMEOUT=30 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=$1
for r in $(seq 0 $(($1-1))); do
  RANK=$r LOCAL_RANK=$r python3 tests/mp_worker.py > gpurun_out/ipc_w$r.out 2> gpurun_out/ipc_w$r.err &
done
wait
for r in $(seq 0 $(($1-1))); do echo "== rank $r"; tail -3 gpurun_out/ipc_w$r.out; grep -v "^  File\|^    " gpurun_out/ipc_w$r.err | tail -4; done
