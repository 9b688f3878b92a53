for rep in 1 2; do
for m in 0 1; do
  echo "## MFFT_COL3S=$m"
  MFFT_COL3S=$m python3 scripts/padprof.py 1024 slab single
done
done
for m in 0 1; do
  echo "## MFFT_COL3S=$m MFFT_COL3=0"
  MFFT_COL3=0 MFFT_COL3S=$m python3 scripts/padprof.py 1024 slab single
done
