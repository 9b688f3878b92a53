"""Find kernels whose global loads the compiler SERIALISED (developer tool, round 4).

    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/opt/rocm/include -S --offload-device-only \
        mpifft4py_amd/csrc/kernels_e_s.hip -o /tmp/kernels_e_s.s
    python scripts/isa_scan.py /tmp/kernels_e_s.s [more .s files]

Per kernel: the number of `global_load ... off` instructions (data loads through a 64-bit address; twiddle tables are read
through a scalar base) and how many of them are followed within two lines by `s_waitcnt vmcnt(0)` -- a load the wave waits
for before it issues the next one.  Kernels with three or more such loads are listed.  What it found and what was done about
it: profiles/r04_serialised_loads.txt."""
import re
import sys


def scan(path):
    s = open(path).read()
    out = []
    for m in re.finditer(r'^(_ZN4mfft\d+mfft_kern\w*INS_\d+(\w+?)INS_4SpecILi(\d+)E\S+):', s, re.M):
        name = m.group(1)
        i = m.end()
        j = s.index('s_endpgm', i)
        lines = s[i:j].split('\n')
        tot = ser = 0
        for k, l in enumerate(lines):
            if 'global_load' in l and ', off' in l:
                tot += 1
                if any('s_waitcnt vmcnt(0)' in x for x in lines[k + 1:k + 3]):
                    ser += 1
        if ser >= 3:
            out.append((m.group(2), int(m.group(3)), name[name.index('Spec'):][:90], tot, ser))
    return out


if __name__ == "__main__":
    for f in sys.argv[1:]:
        for fam, n, spec, tot, ser in scan(f):
            print("%-28s %-9s n=%-5d %s  loads %d, waited for one at a time %d" % (f.split('/')[-1], fam, n, spec, tot, ser))
