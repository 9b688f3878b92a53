#!/bin/bash
# First thing to run on a node with more than one MI355X (nothing in this round could): the multi-process product
# path over REAL RCCL / xGMI -- parity of slab (every pipeline flavour) and pencil transforms at 256^3 and 512^3,
# then the bench at 2, 4 and 8 ranks.  Usage: scripts/multi_gpu_check.sh [max_ranks]
set -u
cd "$(dirname "$0")/.."
MAXR=${1:-8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29610
for P in 2 4 8; do
  [ "$P" -gt "$MAXR" ] && break
  for n in 256 512; do
    echo "== parity: $P ranks, $n^3 (slab pipelines 1/2/4/8, pencils X and Y)"
    MP_N=$n timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 \
      --master-port $((PORT++)) tests/mp_worker_big.py 2>&1 | grep -E "BIG_OK|Error|error|assert" | head -5
  done
  echo "== IPC transport over real links: every pull mode, CU masks, relay striping (tests/mp_worker.py), then a short soak"
  MFFT_TRANSPORT=ipc timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 \
    --master-port $((PORT++)) tests/mp_worker.py 2>&1 | grep -E "MP_OK|Error|error|assert" | head -5
  timeout 600 python scripts/ipc_stress.py $P 1 4 100 256 2>&1 | grep -E "IPC_STRESS|rank" | tail -3
  echo "== bench: $P ranks"
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 \
    --master-port $((PORT++)) bench.py --gpus $P --steps 10 --warmup 3 2>/dev/null | python scripts/show_bench.py
done
