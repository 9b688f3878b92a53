#!/usr/bin/env python3
"""Print README.md's per-size table (ms per pair and fraction of the 8 TB/s roofline) from a sweep file written by
scripts/perf_gate.py / scripts/size_sweep.sh, so that no quoted number is older than the sweep (developer tool).
    python3 scripts/readme_table.py profiles/r05_size_sweep.txt"""
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"^(\d+)\^3 (fp64|fp32) .*?\| ms/pair ([\d.]+) \|.*?frac8TB ([\d.]+)", line)
    if m:
        rows.append((m.group(2) == "fp32", int(m.group(1)), float(m.group(3)), float(m.group(4))))
rows.sort()
head = "| " + " | ".join("%d³%s" % (n, " fp32" if f else "") for f, n, _, _ in rows) + " |"
sep = "|" + "---|" * len(rows)
ms = "| " + " | ".join(("%.2f" if t < 10 else "%.1f") % t for _, _, t, _ in rows) + " |"
fr = "| " + " | ".join("%.2f" % x for _, _, _, x in rows) + " |"
print(head)
print(sep)
print(ms)
print(fr)
