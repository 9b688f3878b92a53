#!/bin/bash
# Soak of the shipped IPC pull modes (0 copy engines, 1 pull kernel, 2 per-peer streams) x exchange flavours, 2 / 4 / 8 processes on one GPU
# (scripts/ipc_stress.py: every iteration's data differs, a stale or early chunk shows).  Round 4.
out=gpurun_out/r04_ipc_soak.txt
: > $out
fail=0; runs=0
for world in 2 4 8; do
for mode in 1 0 2; do
for pipe in 1 4 -4; do
  it=150; [ $world -eq 8 ] && it=40
  [ $world -eq 8 ] && [ $mode -eq 2 ] && continue     # per-peer streams with 8 processes on ONE device: seconds per pair (r03)
  t0=$(date +%s)
  timeout 300 python3 scripts/ipc_stress.py $world $mode $pipe $it 128 > /tmp/soak.log 2>&1
  rc=$?
  runs=$((runs+1)); { [ $rc -ne 0 ] || ! grep -q ": 0 bad blocks" /tmp/soak.log; } && fail=$((fail+1))
  echo "world $world pull $mode pipeline $pipe iterations $it: rc $rc ($(( $(date +%s) - t0 )) s) $(grep IPC_STRESS /tmp/soak.log | tail -1 | cut -c1-140)" >> $out
done
done
done
echo "runs $runs failures $fail" >> $out
cat $out
