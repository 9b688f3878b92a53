#!/bin/bash
# profiles/r03_mixed_radix_15.txt: the plans with both 3 and 5 among their factors (plans.h groups L, M).
# 1. tools/rowcheck: every contiguous-axis kernel of the plans the compiler got wrong, against a host DFT
# 2. single-GPU slab pair at every cube whose side is one of the new lengths
set -u
cd "$(dirname "$0")/.."
echo "== tools/build/rowcheck (built with the row_thread_index workaround of fft_kernels.h)"
tools/build/rowcheck
if [ -x tools/build/rowcheck_nolaunder ]; then
  echo "== the same kernels compiled with -DMFFT_NO_LAUNDER_J (the code as first written)"
  tools/build/rowcheck_nolaunder
fi
echo "== slab R2C pair, fp64 then fp32, one MI355X"
for n in 240 300 360 450 480 600 720 900 960 1200 1440; do
  for p in double single; do
    python bench.py --size $n --precision $p --cpu-baseline off --pencil-extra off --steps 5 2>/dev/null | python scripts/show_bench.py | head -1
  done
done
