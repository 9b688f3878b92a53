#!/bin/bash
# PMC counters of fast and slow processes of the same binary (800^3 slab pair): TLB and L2 / fabric stall counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/placement_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7; do
  timeout 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $O/a$i -- python3 $R/scripts/placement_modes.py 800 5 > $O/a$i.log 2>&1
  timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY TCC_TAG_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ --output-format csv -d $O/b$i -- python3 $R/scripts/placement_modes.py 800 5 > $O/b$i.log 2>&1
done
cd $R
python3 - <<'PY' | tee gpurun_out/r06/placement_pmc.txt
import csv, glob, os, re, collections
O = "gpurun_out/r06/placement_pmc"
for d in sorted(glob.glob(O + "/[ab]*")):
    if not os.path.isdir(d): continue
    log = open(d + ".log").read()
    m = re.search(r"800\^3 pair.*", log)
    fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        print(os.path.basename(d), "no csv", log[-300:].replace("\n", " | ")); continue
    acc = collections.OrderedDict()
    rows = [r for r in csv.DictReader(open(fs[0])) if "ColFft" in r["Kernel_Name"]]
    # the last 5 pairs: 4 strided launches per pair, in order fwd_y fwd_x bwd_x bwd_y
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    print("##", os.path.basename(d), m.group(0) if m else "?")
    for c, v in by.items():
        v = [x for _, x in sorted(v)][-20:]
        per = [sum(v[i::4]) / len(v[i::4]) for i in range(4)]
        print("   %-36s fwd_y %.4g fwd_x %.4g bwd_x %.4g bwd_y %.4g" % (c, *per))
PY
rm -rf $O
