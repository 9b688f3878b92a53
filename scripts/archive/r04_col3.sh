#!/bin/bash
# round 4: 1536 as three 512-point sub-transforms per workgroup (fft_col3.h) against the ColFft plan (MFFT_COL3=0 / 1)
out=gpurun_out/r04_col3_1536.txt
: > $out
python -m pytest tests/test_gpu_stages.py -q -k "1536 or 3072 or 768" 2>&1 | tail -2 >> $out
python -m pytest tests/test_gpu_parity.py -q -k "padded or slab_r2c or dealias or two_thirds" 2>&1 | tail -2 >> $out
for v in "MFFT_COL3=0" "MFFT_COL3=1" "MFFT_COL3=0" "MFFT_COL3=1"; do
  echo "## $v" >> $out
  env $v python3 bench.py --size 1536 --steps 5 --warmup 2 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
  env $v python3 bench.py --size 1536 --precision single --steps 5 --warmup 2 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
  env $v python3 scripts/padprof.py 1024 slab >> $out 2>&1
done
cat $out
