#!/bin/bash
# round 5, session 3: new lengths + fallback + aligned 3/2-rule route (tests), padded pair A/B, radix-7 gate, headline check
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "padded or beyond_the_radix or three_halves" > gpurun_out/r05_t3a.log 2>&1; tail -4 gpurun_out/r05_t3a.log
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q > gpurun_out/r05_t3b.log 2>&1; tail -4 gpurun_out/r05_t3b.log
out=gpurun_out/r05_pad_align_ab.txt
: > $out
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 2" "0 2"; do
  set -- $cfg
  echo "== MFFT_PAD_ALIGN=$1 MFFT_COL3S=$2 (rep $rep)" >> $out
  MFFT_PAD_ALIGN=$1 MFFT_COL3S=$2 python3 scripts/padprof.py 1024 slab double >> $out 2>&1
done
done
for cfg in "1 1" "0 1"; do
  set -- $cfg
  echo "== MFFT_PAD_ALIGN=$1 single" >> $out
  MFFT_PAD_ALIGN=$1 python3 scripts/padprof.py 1024 slab single >> $out 2>&1
  echo "== MFFT_PAD_ALIGN=$1 512 double" >> $out
  MFFT_PAD_ALIGN=$1 python3 scripts/padprof.py 512 slab double >> $out 2>&1
  echo "== MFFT_PAD_ALIGN=$1 768 double (padded 1152)" >> $out
  MFFT_PAD_ALIGN=$1 python3 scripts/padprof.py 768 slab double >> $out 2>&1
done
cat $out
python3 scripts/perf_gate.py --baseline profiles/r04_radix7_sweep.txt --out gpurun_out/r05_radix7_gate2.txt > gpurun_out/r05_radix7_gate2.log 2>&1
tail -12 gpurun_out/r05_radix7_gate2.log
python3 scripts/perf_gate.py --sizes 512 1024 1536 2048 --out gpurun_out/r05_gate_headline.txt > gpurun_out/r05_gate_headline.log 2>&1
tail -14 gpurun_out/r05_gate_headline.log
