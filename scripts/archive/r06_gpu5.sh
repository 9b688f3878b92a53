#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06; out=gpurun_out/r06/dns_batch.txt
: > $out
for mb in 80 160 320 640 1536 6000; do
  echo "## MFFT_NLZ_BATCH_MB=$mb" >> $out
  MFFT_NLZ_VARIANT=11 MFFT_NLZ_BATCH_MB=$mb timeout 300 python examples/spectral_dns_device.py --M 9 --steps 3 --stages >> $out 2>&1
done
echo "## align=0" >> $out
MFFT_NLZ_VARIANT=11 MFFT_NLZ_ALIGN=0 timeout 300 python examples/spectral_dns_device.py --M 9 --steps 3 --stages >> $out 2>&1
cat $out
