#!/usr/bin/env python3
"""Stage-by-stage performance gate (developer tool; runs on the GPU box).

    python3 scripts/perf_gate.py                      # sweep of HEAD against the PREVIOUS round's committed sweep
    python3 scripts/perf_gate.py --baseline profiles/r04_radix7_sweep.txt --sizes 896 1792
    python3 scripts/perf_gate.py --compare gpurun_out/size_sweep.txt      # no GPU: compare two sweep files

Runs `bench.py --size n --precision p` (a fresh process each) for every (n, p) line of the baseline sweep (the format
scripts/size_sweep.sh writes: one line per mesh with the six stage times of the forward + inverse pair), writes the new
sweep (gpurun_out/size_sweep.txt) and compares every stage with the baseline's.

What a failure means (round 6).  The same binary lands in one of (at least) two PLACEMENT MODES per process: the driver
gives an allocation other physical memory every time, the strided passes over it then meet the DRAM channels more or less
evenly (same TLB misses, 1.5 x the DRAM credit stalls and +10 % read latency in the slow mode: profiles/r06_placement_pmc.txt),
and HIP offers no way to steer it (profiles/r02_placement_reroll.txt).  Measured spread (profiles/r06_placement_pmc.txt,
r06_perf_gate.log): 3 - 8 % on single strided stages at 768^3 / 800^3 (7 of 10 processes slow), up to 15 % elsewhere (1000^3
y passes 3.45 or 3.82 ms, the same kernel since round 4; 896^3 bwd_y 2.15 or 2.46; 480^3 bwd_x 0.33 or 0.38), up to 8 % on a pair
(1000^3: 18.5 - 20.7 ms over six sessions).  A single-run baseline may be a fast sample, so:
  * a mesh whose first run is inside --tol (5 %) on every stage passes at once;
  * otherwise the mesh is measured again in --fresh (3) more fresh processes and the MEDIAN of every stage (and of the
    pair) over all runs counts;
  * a REGRESSION is a median stage more than --stage-tol (20 %: the mode spread plus --tol) slower than the baseline, or a
    median pair more than --pair-tol (10 %) slower.  Round 4's loss (2 x on two kernels, +22 % on the pair) and anything like
    it is outside both; what is inside cannot be told from a placement mode by timing at all (that takes the counters).  A
    tree with no kernel change passes twice in a row (profiles/r06_perf_gate.log).
Stages below 0.05 ms (launch-bound meshes) are not judged.  The baseline is the newest committed profiles/rNN_size_sweep.txt
of an EARLIER round than the tree's own (profiles/ROUND names it), never the sweep the tree itself produced."""
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
    """The newest committed sweep of an earlier round than this tree's (profiles/ROUND holds the tree's round number)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_size_sweep.txt")))
    try:
        cur = int(open(os.path.join(ROOT, "profiles", "ROUND")).read().split()[0])
    except (OSError, ValueError, IndexError):
        cur = 10 ** 6
    files = [f for f in files if int(os.path.basename(f)[1:3]) < cur]
    if not files:
        sys.exit("no profiles/r*_size_sweep.txt of an earlier round to compare with")
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


def slower(base, new, tol, floor, pair_tol=None):
    bad = []
    for st, b in base[1].items():
        v = new[1].get(st)
        if v is None or b < floor:
            continue
        if v > b * (1 + tol) + 0.006:             # the sweep files carry two decimals: half a unit of rounding on each side
            bad.append((st, b, v))
    if pair_tol is not None and base[0] >= 6 * floor and new[0] > base[0] * (1 + pair_tol) + 0.006:
        bad.append(("pair", base[0], new[0]))
    return bad


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


EXTRA = [(432, "fp64"), (864, "fp64"), (1728, "fp64"), (864, "fp32"), (1728, "fp32"), (560, "fp32"), (1120, "fp32"),
         (648, "fp64"), (1008, "fp64"), (1080, "fp64"), (1296, "fp64"), (1080, "fp32")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--baseline", default=None)
    ap.add_argument("--compare", default=None, help="compare this sweep file with the baseline instead of measuring")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "size_sweep.txt"))
    ap.add_argument("--sizes", type=int, nargs="*", default=None)
    ap.add_argument("--precisions", nargs="*", default=None, choices=["fp64", "fp32"])
    ap.add_argument("--tol", type=float, default=0.05, help="first run: every stage within this of the baseline passes at once")
    ap.add_argument("--stage-tol", type=float, default=0.20, help="median of the fresh processes: a stage beyond this is a regression")
    ap.add_argument("--pair-tol", type=float, default=0.10, help="... and so is a pair beyond this")
    ap.add_argument("--floor", type=float, default=0.05)
    ap.add_argument("--fresh", type=int, default=3, help="fresh processes a flagged mesh is measured again in")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    base_path = a.baseline or newest_baseline()
    base = parse(base_path)
    keys = [k for k in base if (not a.sizes or k[0] in a.sizes) and (not a.precisions or k[1] in a.precisions)]
    remeasured = {}
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
            if slower(base[(n, prec)], (ms, stages), a.tol, a.floor, a.tol):
                runs = [(ms, stages, d)]
                for _ in range(a.fresh):
                    ms2, st2, d2 = measure(n, prec, a.steps if n < 2000 else max(3, a.steps // 2))
                    if ms2 is not None:
                        runs.append((ms2, st2, d2))
                ms = median([r[0] for r in runs])
                stages = {k: median([r[1].get(k, v) for r in runs]) for k, v in stages.items()}
                remeasured[(n, prec)] = len(runs)
            new[(n, prec)] = (ms, stages)
            lines.append(fmt(n, prec, ms, stages, d))
            print(lines[-1], flush=True)
        # meshes the baseline does not have yet (plans added this round): measured once and recorded, not judged
        for (n, prec) in EXTRA:
            if (n, prec) in base or (a.sizes and n not in a.sizes) or (a.precisions and prec not in a.precisions):
                continue
            ms, stages, d = measure(n, prec, a.steps if n < 2000 else max(3, a.steps // 2))
            if ms is not None:
                lines.append(fmt(n, prec, ms, stages, d))
                print(lines[-1] + "   (new this round: no baseline)", flush=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    print("\nbaseline %s\n%-14s %9s %9s %7s   median stage > %.0f %% or median pair > %.0f %% slower (a first run inside %.0f %% passes at once)" % (
        os.path.relpath(base_path, ROOT), "mesh", "base ms", "new ms", "ratio", 100 * a.stage_tol, 100 * a.pair_tol, 100 * a.tol))
    failed = 0
    for k in keys:
        if k not in new:
            continue
        bad = slower(base[k], new[k], a.stage_tol, a.floor, a.pair_tol)
        failed += bool(bad)
        note = " (median of %d fresh processes)" % remeasured[k] if k in remeasured else ""
        print("%-14s %9.2f %9.2f %7.3f   %s%s" % ("%d^3 %s" % k, base[k][0], new[k][0], new[k][0] / base[k][0],
                                                   ", ".join("%s %.2f -> %.2f" % b for b in bad) or "-", note))
    print("\nperf gate: %s" % ("%d mesh(es) REGRESSED" % failed if failed else "ok"))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
