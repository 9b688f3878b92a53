#!/usr/bin/env python3
"""Stage-by-stage performance gate (developer tool; runs on the GPU box).

    python3 scripts/perf_gate.py                      # sweep of HEAD against the newest committed profiles/r*_size_sweep.txt
    python3 scripts/perf_gate.py --baseline profiles/r04_radix7_sweep.txt --sizes 896 1792
    python3 scripts/perf_gate.py --compare gpurun_out/size_sweep.txt      # no GPU: compare two sweep files

Runs `bench.py --size n --precision p` for every (n, p) line of the baseline sweep (the format scripts/size_sweep.sh
writes: one line per mesh with the six stage times of the forward + inverse pair), writes the new sweep next to it
(gpurun_out/size_sweep.txt) and compares every stage with the baseline's.  A stage more than --tol (5 %) slower is
measured again (up to --retries more runs of that mesh, the fastest time of each stage counts: the boxes of the pool differ
by 1 - 3 % and a single run can land on a slow placement); what is still slower is a REGRESSION: listed, exit status 1.
Stages below 0.05 ms (launch-bound meshes) are not judged.  The end-of-round script runs this before the docs quote numbers
(the final binary of round 4 had lost 2 x on two kernels and nothing noticed)."""
import argparse
import ast
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^(\d+)\^3 (fp64|fp32) .*?\| ms/pair ([\d.]+) \|.*?(\{.*?\})")


def parse(path):
    """{(n, precision): (ms_per_pair, {stage: ms})}; a mesh measured twice keeps the last line."""
    out = {}
    for line in open(path):
        m = LINE.match(line.strip())
        if m:
            out[(int(m.group(1)), m.group(2))] = (float(m.group(3)), ast.literal_eval(m.group(4)))
    return out


def newest_baseline():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_size_sweep.txt")))
    if not files:
        sys.exit("no profiles/r*_size_sweep.txt to compare with")
    return files[-1]


def measure(n, prec, steps):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", str(n), "--precision",
           "double" if prec == "fp64" else "single", "--steps", str(steps), "--warmup", "3", "--cpu-baseline", "off",
           "--pencil-extra", "off"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT)
    for line in p.stdout.decode().splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            return d["ms_per_step"], dict(d["config"]["stage_ms"]), d
    return None, None, None


def fmt(n, prec, ms, stages, d):
    w = d["config"]["workload"][:40] if d else "%d^3 %s slab R2C forward+inverse" % (n, prec)
    return "%s | ms/pair %s | pairs/s %s | frac8TB %s %s rt %s" % (
        w, round(ms, 2), round(1e3 / ms, 1), round(d["config"]["whole_path_frac_of_8TBs"], 3) if d else "-",
        {k: round(v, 2) for k, v in sorted(stages.items())}, "%.1e" % d["config"]["roundtrip_rel_l2"] if d else "-")


def slower(base, new, tol, floor):
    bad = []
    for st, b in base[1].items():
        v = new[1].get(st)
        if v is None or b < floor:
            continue
        if v > b * (1 + tol) + 0.006:             # the sweep files carry two decimals: half a unit of rounding on each side
            bad.append((st, b, v))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--baseline", default=None)
    ap.add_argument("--compare", default=None, help="compare this sweep file with the baseline instead of measuring")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "size_sweep.txt"))
    ap.add_argument("--sizes", type=int, nargs="*", default=None)
    ap.add_argument("--precisions", nargs="*", default=None, choices=["fp64", "fp32"])
    ap.add_argument("--tol", type=float, default=0.05)
    ap.add_argument("--floor", type=float, default=0.05)
    ap.add_argument("--retries", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    base_path = a.baseline or newest_baseline()
    base = parse(base_path)
    keys = [k for k in base if (not a.sizes or k[0] in a.sizes) and (not a.precisions or k[1] in a.precisions)]
    if a.compare:
        new = parse(a.compare)
    else:
        new = {}
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        lines = []
        for (n, prec) in keys:
            ms, stages, d = measure(n, prec, a.steps if n < 2000 else max(3, a.steps // 2))
            if ms is None:
                print("bench.py failed for %d^3 %s" % (n, prec))
                new[(n, prec)] = (float("inf"), {st: float("inf") for st in base[(n, prec)][1]})
                continue
            tries = 0
            while slower(base[(n, prec)], (ms, stages), a.tol, a.floor) and tries < a.retries:
                ms2, st2, d2 = measure(n, prec, a.steps)
                tries += 1
                if ms2 is None:
                    break
                stages = {k: min(v, st2.get(k, v)) for k, v in stages.items()}
                if ms2 < ms:
                    ms, d = ms2, d2
            new[(n, prec)] = (ms, stages)
            lines.append(fmt(n, prec, ms, stages, d))
            print(lines[-1], flush=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    print("\nbaseline %s\n%-14s %9s %9s %7s   stages more than %.0f %% slower" % (
        os.path.relpath(base_path, ROOT), "mesh", "base ms", "new ms", "ratio", 100 * a.tol))
    failed = 0
    for k in keys:
        if k not in new:
            continue
        bad = slower(base[k], new[k], a.tol, a.floor)
        failed += bool(bad)
        print("%-14s %9.2f %9.2f %7.3f   %s" % ("%d^3 %s" % k, base[k][0], new[k][0], new[k][0] / base[k][0],
                                                 ", ".join("%s %.2f -> %.2f" % b for b in bad) or "-"))
    print("\nperf gate: %s" % ("%d mesh(es) REGRESSED" % failed if failed else "ok"))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
