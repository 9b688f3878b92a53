#!/bin/bash
# round 4: A/B of the x pass behind an exchange, the one-rank out-of-place forward, config 5
out=gpurun_out/r04_xpass_ab.txt
: > $out
run() { echo "## $*" >> $out; env "$@" >> $out 2>&1; }
for v in "" "MFFT_NO_XPAD=1" "MFFT_NO_XPAD=1 MFFT_XPASS_INPLACE=1"; do
  run $v python3 scripts/xpass_ab.py 1024 8 c2cX single
  run $v python3 scripts/xpass_ab.py 1024 8 c2cY single
  run $v python3 scripts/xpass_ab.py 1024 4 slabc2c single
  run $v python3 scripts/xpass_ab.py 1024 2 slabc2c double
done
for v in "" "MFFT_XPASS_INPLACE=1"; do
  run $v python3 scripts/xpass_ab.py 1024 2 slab double
  run $v python3 scripts/xpass_ab.py 1024 8 slab double
  run $v python3 scripts/xpass_ab.py 1024 8 pencilX double
  run $v python3 scripts/xpass_ab.py 1024 8 pencilY double
done
echo "## one rank: forward y / x passes in place (default) against out of place (MFFT_FWD_OOP=1)" >> $out
for v in "MFFT_FWD_OOP=0" "MFFT_FWD_OOP=1" "MFFT_FWD_OOP=0" "MFFT_FWD_OOP=1"; do
  for n in 512 1024; do
    echo "# $v n=$n" >> $out
    env $v python3 bench.py --size $n --steps 10 --warmup 3 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
  done
done
echo "# fp32 1024, 2048" >> $out
for v in "MFFT_FWD_OOP=0" "MFFT_FWD_OOP=1"; do
  for n in 1024 2048; do
    echo "# $v n=$n single" >> $out
    env $v python3 bench.py --size $n --precision single --steps 5 --warmup 2 --cpu-baseline off --pencil-extra off 2>/dev/null | python3 scripts/show_bench.py >> $out
  done
done
echo "## c2cprof 2048 single" >> $out
python3 scripts/c2cprof.py 2048 single >> $out 2>&1
echo "## config 5 at full size, stage times" >> $out
for v in "" "MFFT_NO_XPAD=1 MFFT_XPASS_INPLACE=1"; do
  echo "# [$v]" >> $out
  env CONFIG5_STAGES=1 $v python3 scripts/config5_full.py 2048 8 >> $out 2>&1
done
cat $out
