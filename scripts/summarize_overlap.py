"""Turn the per-rank rocprofv3 traces of scripts/overlap_trace.sh into a table: for every rank, and for every piece of
the pipelined exchange of the LAST traced transform pair, how long the exchange ran and for how much of that time a
transform kernel OF THE SAME RANK was executing (intersection of the intervals on the device timeline).

    python scripts/summarize_overlap.py gpurun_out/overlap_<tag> [> profiles/r03_overlap_<tag>.txt]
"""
import csv
import glob
import os
import re
import sys


def load(rank_dir):
    ev = []      # (start_ns, end_ns, kind, name)
    for path in glob.glob(os.path.join(rank_dir, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row.get("Kernel_Name", "")
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            if "ipc_pull_kernel" in name:
                kind = "xchg"
            elif "mfft_kern" in name or "ColFft" in name or "R2CFft" in name or "C2RFft" in name:
                kind = "fft"
            else:
                kind = "other"
            short = re.sub(r"^.*mfft::(ColFft|R2CFft|C2RFft|RowFft).*$", r"\1", name)
            ev.append((s, e, kind, short if len(short) < 40 else short[:40]))
    for path in glob.glob(os.path.join(rank_dir, "**", "*memory_copy_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "DEVICE_TO_DEVICE" not in row.get("Direction", "").upper().replace("MEMORY_COPY_", ""):
                continue
            ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), "xchg", "copy"))
    ev.sort()
    return ev


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def intersect(a, b):
    """total length of the intersection of interval [a0, a1) with the merged interval list b"""
    tot = 0
    for s, e in b:
        lo, hi = max(a[0], s), min(a[1], e)
        if hi > lo:
            tot += hi - lo
    return tot


def by_thread(root):
    """One trace of ONE process whose ranks are host threads (scripts/overlap_local.py): kernels grouped by the
    dispatching thread; a rank's exchange = the device-copy kernels its thread issued."""
    per = {}
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row.get("Kernel_Name", "")
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            kind = "xchg" if "copyBuffer" in name else "fft" if "mfft_kern" in name else None
            if kind:
                per.setdefault(row["Thread_Id"], {"xchg": [], "fft": []})[kind].append((s, e))
    print("# %s: kernels of %d dispatching threads" % (root, len(per)))
    tx = tov = 0
    for i, (tid, d) in enumerate(sorted(per.items(), key=lambda kv: kv[0])):
        if not d["xchg"] or not d["fft"]:
            continue
        fft = merged(d["fft"])
        t_x = sum(e - s for s, e in d["xchg"])
        t_ov = sum(intersect(iv, fft) for iv in d["xchg"])
        tx += t_x; tov += t_ov
        print("thread %s: %d copies %.2f ms, %d transform kernels %.2f ms, copy time beside a transform kernel of the same "
              "rank: %.2f ms = %.0f %%" % (tid, len(d["xchg"]), t_x / 1e6, len(d["fft"]), sum(e - s for s, e in fft) / 1e6,
                                          t_ov / 1e6, 100.0 * t_ov / max(t_x, 1)))
    if tx:
        print("# all ranks: %.0f %% of the copy time ran beside a transform kernel of the same rank" % (100.0 * tov / tx))


def main():
    if sys.argv[1] == "--by-thread":
        return by_thread(sys.argv[2])
    root = sys.argv[1]
    ranks = sorted(glob.glob(os.path.join(root, "rank*/")), key=lambda p: int(re.search(r"rank(\d+)", p).group(1)))
    print("# %s: %d rank traces" % (root, len(ranks)))
    for log in sorted(glob.glob(os.path.join(root, "rank*.log"))):
        for line in open(log):
            if "ms per pair" in line:
                print("# " + line.strip())
    tot_x = tot_ov = tot_f = 0
    for rd in ranks:
        r = int(re.search(r"rank(\d+)", rd).group(1))
        ev = load(rd)
        fft = merged([(s, e) for s, e, k, _ in ev if k == "fft"])
        xch = [(s, e, nm) for s, e, k, nm in ev if k == "xchg"]
        if not xch or not fft:
            print("rank %d: no exchange / transform records" % r)
            continue
        t_x = sum(e - s for s, e, _ in xch)
        t_f = sum(e - s for s, e in fft)
        t_ov = sum(intersect((s, e), fft) for s, e, _ in xch)
        tot_x += t_x; tot_ov += t_ov; tot_f += t_f
        print("rank %d: whole trace: exchange %.2f ms in %d operations, transforms %.2f ms, exchange time during which a "
              "transform kernel of this rank was running: %.2f ms = %.0f %%"
              % (r, t_x / 1e6, len(xch), t_f / 1e6, t_ov / 1e6, 100.0 * t_ov / max(t_x, 1)))
        if r == 0:
            # the last pair in detail: the final 2 * pieces exchange operations (kernel mode: one per piece)
            npc = int(os.environ.get("OVERLAP_PIECES", "8"))
            last = xch[-npc:]
            t0 = last[0][0]
            print("   rank 0, last pair, per exchange piece (times relative to the first one's start, ms):")
            print("   %-6s %9s %9s %9s %12s" % ("piece", "start", "end", "length", "beside fft"))
            for i, (s, e, nm) in enumerate(last):
                print("   %-6d %9.3f %9.3f %9.3f %9.3f ms" % (i, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, intersect((s, e), fft) / 1e6))
            print("   transform kernels of rank 0 in that window:")
            for s, e, k, nm in ev:
                if k == "fft" and e >= t0 - 2e6 and s <= last[-1][1] + 2e6:
                    print("     %-10s %9.3f .. %9.3f" % (nm, (s - t0) / 1e6, (e - t0) / 1e6))
    if tot_x:
        print("# all ranks: %.0f %% of the exchange time ran beside a transform kernel of the same rank "
              "(exchange %.1f ms, transforms %.1f ms)" % (100.0 * tot_ov / tot_x, tot_x / 1e6, tot_f / 1e6))


if __name__ == "__main__":
    main()
