"""Turn rocprofv3 output directories (kernel trace + separate FETCH_SIZE / WRITE_SIZE
--pmc passes of the same bench.py command) into the small files kept under profiles/.

    python scripts/summarize_profiles.py <tag> <trace_dir> <fetch_dir> <write_dir> "<workload>"

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in
KiB and on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, "HBM"); calibrated here on the r2c kernel, whose read volume is
exactly the real array.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    m = re.search(r"mfft::(ColFft|RowFft|R2CFft|C2RFft)<mfft::Spec<([\d, ]+)>, (\w+), (\d+), (\w+)([\w, ]*)>", name)
    if not m:
        return re.sub(r"\(.*", "", name)[:60]
    fam, spec, prec, tile, flag, rest = m.groups()
    extra = ""
    if fam in ("ColFft", "RowFft"):
        extra = " inv" if flag == "true" else " fwd"
    tail = [t.strip() for t in rest.split(",") if t.strip()]
    if fam == "ColFft":                                               # TWLDS, SPLIT, VEC, NT, PAD
        if len(tail) >= 4 and tail[3] == "true":
            extra += " nt"
        if len(tail) >= 5 and tail[4] != "0":
            extra += " pad%s" % tail[4]
    elif fam in ("R2CFft", "C2RFft"):                                 # (TWLDS), LIMIT, CHUNK, SPLIT
        for i, nm in enumerate(("limit", "chunk", "split")):
            if len(tail) > i and tail[i] == "true":
                extra += " " + nm
    elif fam == "RowFft":                                             # (INV), TWLDS, CHUNK, SPLIT
        for i, nm in ((1, "chunk"), (2, "split")):
            if len(tail) > i and tail[i] == "true":
                extra += " " + nm
    return "%s n=%s %s tile=%s%s" % (fam, spec.replace(", ", "x"), prec, tile, extra)


def pmc(dirname):
    f = glob.glob(os.path.join(dirname, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def sq_counters(tag, dirs, out_dir):
    """Per-kernel averages of the SQ / GRBM counters of one or more --pmc passes -> <tag>_pmc_sq_counters.json"""
    res = collections.defaultdict(dict)
    for d in dirs:
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            res[k][c] = sum(v) / len(v)
    for k, c in res.items():
        if c.get("SQ_WAVE_CYCLES"):
            for name in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if name in c:
                    c[name + "_share_of_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_WAVES"):
            for name in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if name in c:
                    c[name + "_per_wave"] = c[name] / c["SQ_WAVES"]
    with open(os.path.join(out_dir, "%s_pmc_sq_counters.json" % tag), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    return res


def main():
    if sys.argv[1] == "sq":
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
        print(json.dumps(sq_counters(sys.argv[2], sys.argv[3:], out_dir), indent=1, sort_keys=True))
        return
    tag, trace_dir, fetch_dir, write_dir, workload = sys.argv[1:6]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    stats = glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(out_dir, "%s_kernel_stats.csv" % tag))
    fetch, nf = pmc(fetch_dir)
    write, _ = pmc(write_dir)
    kern = {}
    for row in csv.DictReader(open(stats)):
        kern[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6}
    res = {"workload": workload, "kernels": {}}
    for k in fetch:
        if k not in write:
            continue
        res["kernels"][k] = {
            "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write[k], "dispatches_sampled": nf[k],
            "hbm_bytes_per_launch": (2.0 * fetch[k] + write[k]) * 1024.0,
            "avg_ms_kernel_trace": kern.get(k, {}).get("avg_ms"),
            "calls_kernel_trace": kern.get(k, {}).get("calls"),
        }
        ms = kern.get(k, {}).get("avg_ms")
        if ms:      # HBM GB/s of the kernel: counter bytes per launch over its average duration in the kernel trace
            res["kernels"][k]["hbm_GBs"] = res["kernels"][k]["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9
            res["kernels"][k]["frac_of_8TBs"] = res["kernels"][k]["hbm_GBs"] / 8000.0
    with open(os.path.join(out_dir, "%s_pmc_traffic.json" % tag), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
