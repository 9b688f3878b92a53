#!/bin/bash
# HBM traffic of the stages of one 3/2-rule ifftn + fftn pair (rocprofv3 PMC, separate passes), compact (MFFT_PAD_ALIGN=0) and line-aligned (=1) intermediates.  Round 5.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pad_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export MFFT_PAD_ALIGN=$m
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch$m -- python3 $R/scripts/padprof.py 1024 slab > $O/fetch$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write$m -- python3 $R/scripts/padprof.py 1024 slab > $O/write$m.log 2>&1
done
cd $R
python3 - <<'PY' | tee gpurun_out/r05_pad_pmc.txt
import csv, glob, os, re
O = "gpurun_out/pad_pmc"
for m in (0, 1):
    res = {}
    for what in ("fetch", "write"):
        f = glob.glob(os.path.join(O, "%s%d" % (what, m), "**", "*_counter_collection.csv"), recursive=True)[0]
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        rows = [r for r in rows if "mfft" in r["Kernel_Name"] and "fill" not in r["Kernel_Name"]]
        last = rows[-6:]                       # the last pair: ifftn (x, y, z) + fftn (z, y, x)
        for i, r in enumerate(last):
            nm = re.sub(r"void mfft::mfft_kern(_occ)?<mfft::", "", r["Kernel_Name"])[:70]
            res.setdefault(i, {"name": nm})[what] = float(r["Counter_Value"])
    print("## MFFT_PAD_ALIGN=%d: the six kernels of the last pair; HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB" % m)
    for i in sorted(res):
        d = res[i]
        print("  %-72s fetch %8.2f GB  write %8.2f GB  total %8.2f GB" % (d["name"], 2 * d["fetch"] * 1024 / 1e9, d["write"] * 1024 / 1e9,
                                                                       (2 * d["fetch"] + d["write"]) * 1024 / 1e9))
PY
rm -rf $O
