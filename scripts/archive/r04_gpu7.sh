#!/bin/bash
python -m pytest tests -m gpu -q > gpurun_out/r04_gputests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04_gputests.log
tail -6 gpurun_out/r04_gputests.log
out=gpurun_out/r04_zpitch_ab.txt
: > $out
for v in "" "MFFT_NO_ZPITCH=1" "" "MFFT_NO_ZPITCH=1"; do
  echo "## [$v]" >> $out
  env $v python3 scripts/xpass_ab.py 1024 8 pencilX double >> $out 2>&1
  env $v python3 scripts/xpass_ab.py 1024 8 pencilY double >> $out 2>&1
done
cat $out
