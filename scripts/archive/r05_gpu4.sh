#!/bin/bash
# round 5, session 4: why the aligned and the compact 3/2-rule routes differ in bits; inverse flavours of the aligned route;
# config 5 with the x-aligned z-row pitch; tests of this session's changes
for cfg in "32 64 128 single" "32 64 128 double" "64 32 1024 single" "20 12 44 double"; do
  echo "== $cfg"; python3 scripts/pad_align_diff.py $cfg 2>&1 | tail -12
done > gpurun_out/r05_pad_align_diff.txt 2>&1
cat gpurun_out/r05_pad_align_diff.txt
out=gpurun_out/r05_pad_align_inv.txt
: > $out
for rep in 1 2; do
for inv in 1 2 3; do
  echo "== MFFT_PAD_ALIGN=1 MFFT_PAD_ALIGN_INV=$inv (rep $rep)" >> $out
  MFFT_PAD_ALIGN=1 MFFT_PAD_ALIGN_INV=$inv python3 scripts/padprof.py 1024 slab double >> $out 2>&1
done
echo "== MFFT_PAD_ALIGN=0 (rep $rep)" >> $out
MFFT_PAD_ALIGN=0 python3 scripts/padprof.py 1024 slab double >> $out 2>&1
done
for inv in 1 3; do
  echo "== 1536 padded 2304: MFFT_PAD_ALIGN_INV=$inv" >> $out
  MFFT_PAD_ALIGN_INV=$inv python3 scripts/padprof.py 1536 slab double >> $out 2>&1
done
echo "== 1536 MFFT_PAD_ALIGN=0" >> $out
MFFT_PAD_ALIGN=0 python3 scripts/padprof.py 1536 slab double >> $out 2>&1
cat $out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "config5 or padded or beyond_the_radix or pencil_x_exchange or pencil_c2c" > gpurun_out/r05_t4a.log 2>&1; tail -4 gpurun_out/r05_t4a.log
timeout 600 python3 -m pytest tests/test_gpu_stages.py -x -q -k "unsupported or scratch" > gpurun_out/r05_t4b.log 2>&1; tail -3 gpurun_out/r05_t4b.log
echo "== config 5, 8 virtual ranks: z-row pitch on" > gpurun_out/r05_config5.txt
timeout 900 python3 scripts/config5_full.py >> gpurun_out/r05_config5.txt 2>&1
echo "== config 5: MFFT_NO_ZPITCH=1" >> gpurun_out/r05_config5.txt
MFFT_NO_ZPITCH=1 timeout 900 python3 scripts/config5_full.py >> gpurun_out/r05_config5.txt 2>&1
grep -v "^$" gpurun_out/r05_config5.txt | tail -30
