#!/usr/bin/env python3
"""Registers and scratch of every gfx950 kernel in the built objects (developer tool, no GPU needed).

    python3 scripts/kernel_regs.py [build dir] > regs.tsv          # one line per kernel: vgpr agpr scratch lds symbol
    python3 scripts/kernel_regs.py --diff old.tsv new.tsv          # kernels whose numbers changed, worst first

Reads the code-object metadata (clang-offload-bundler --unbundle, llvm-readelf --notes) of mpifft4py_amd/csrc/build/*.o.
Why it exists: in round 4 an edit that did not concern them moved the c2r kernels of the 7 * 2^a plans from 248 to 262
VGPRs -- from two waves per SIMD to one, 2 x the time -- and no test or benchmark noticed (profiles/r05_radix7_c2r_bisect.txt).
Run before and after a change to a kernel header; `--diff` lists what moved across an occupancy step (512 / n waves:
128, 168, 256 VGPRs) or started to spill."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, stderr=subprocess.DEVNULL)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, check=True).stdout.decode()
    out = []
    for blk in notes.split("  - .agpr_count:")[1:]:
        f = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk))
        if "name" in f:
            out.append((int(f.get("vgpr_count", 0)), int(f.get("agpr_count", 0)), int(f.get("private_segment_fixed_size", 0)),
                        int(f.get("group_segment_fixed_size", 0)), f["name"]))
    return out


def waves(v, a):
    tot = (v + a + 7) // 8 * 8
    return max(1, min(8, 512 // max(tot, 1)))


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--diff":
        def load(p):
            return {l.split("\t")[4].strip(): tuple(int(x) for x in l.split("\t")[:4]) for l in open(p) if l.strip()}
        old, new = load(sys.argv[2]), load(sys.argv[3])
        rows = []
        for k in sorted(set(old) & set(new)):
            o, n = old[k], new[k]
            if o[:3] != n[:3]:
                rows.append((waves(n[0], n[1]) - waves(o[0], o[1]), n[2] - o[2], k, o, n))
        rows.sort()
        names = subprocess.run(["c++filt"], input="\n".join(r[2] for r in rows), stdout=subprocess.PIPE, text=True).stdout.split("\n")
        for (dw, ds, k, o, n), nm in zip(rows, names):
            print("waves/SIMD %d -> %d  vgpr+agpr %d+%d -> %d+%d  scratch %d -> %d  %s" % (
                waves(o[0], o[1]), waves(n[0], n[1]), o[0], o[1], n[0], n[1], o[2], n[2], nm[:170]))
        print("# %d kernels changed (of %d common; %d only old, %d only new)" % (len(rows), len(set(old) & set(new)),
                                                                                  len(set(old) - set(new)), len(set(new) - set(old))))
        return
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mpifft4py_amd", "csrc", "build")
    for obj in sorted(glob.glob(os.path.join(build, "*.o"))):
        for v, a, s, l, name in kernels_of(obj):
            print("%d\t%d\t%d\t%d\t%s" % (v, a, s, l, name))


if __name__ == "__main__":
    main()
