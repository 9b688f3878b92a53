#!/bin/bash
# Kernel + memory-copy timeline of every rank of a pipelined slab transform, one rocprofv3 per rank (the program after
# `--` is python3 itself; nothing re-executes after the GPU is initialised):
#   scripts/overlap_trace.sh <tag> <world> <size> <pipeline> [pull mode 0|1|2] [comm_cus]
# Output: gpurun_out/overlap_<tag>/rank<r>/...; scripts/summarize_overlap.py turns it into profiles/r03_overlap*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; W=$2; SIZE=$3; PIPE=$4; PULL=${5:-1}; CUS=${6:-0}
O=$R/gpurun_out/overlap_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PORT=$((20000 + RANDOM % 20000))
pids=()
for r in $(seq 0 $((W - 1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$W LOCAL_WORLD_SIZE=$W MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT MFFT_TRANSPORT=ipc \
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/rank$r -o trace -- \
    python3 $R/scripts/overlap_worker.py --size $SIZE --pipeline $PIPE --pull $PULL --comm-cus $CUS > $O/rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
cd $R
grep -h "ms per pair" $O/rank*.log
find $O -name "*.db" -delete
echo "overlap_trace $TAG rc=$rc"; du -sh $O
