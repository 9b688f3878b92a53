"""The compiled kernels keep a thread's global loads in flight TOGETHER (round 4).

hipcc turned `cond ? load : 0` and per-row `if (columns exist) load` into branches with `s_waitcnt vmcnt(0)` inside:
every load a round trip to HBM of its own (single precision: the column-limited and z-chunked c2r kernels, ColFft3;
profiles/r04_serialised_loads.txt).  The sources now avoid the pattern (fft_core.h keep_bits, one branch around a whole
load phase); this test reads the ISA hipcc produces here and fails if it comes back in the kernels of the BASELINE
configurations (group B, double precision: 512 / 1024) or in the kernels that had it (group E, single precision).
No GPU needed: hipcc cross-compiles."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = "/opt/rocm/bin/hipcc"


def _asm(unit, tmp):
    out = os.path.join(tmp, unit + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I/opt/rocm/include", "-S", "--offload-device-only",
           os.path.join(ROOT, "mpifft4py_amd", "csrc", unit + ".hip"), "-o", out]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_serialised_loads_in_the_baseline_and_the_repaired_kernels(tmp_path):
    import isa_scan
    with ThreadPoolExecutor(2) as ex:
        b_d, e_s = ex.map(lambda u: _asm(u, str(tmp_path)), ["kernels_b_d", "kernels_e_s"])
    # group B, double precision: every kernel family of 512 / 1024 except chirp-z, and except the z-chunked / column-limited
    # c2r of complex length 1024 (real 2048), whose four cached pre-pass twiddle loads are waited for (not a BASELINE path)
    bad = [r for r in isa_scan.scan(b_d) if not r[0].endswith("Z") and not (r[0] == "C2RFft" and r[1] == 1024 and r[4] <= 4)]
    assert not bad, bad
    # group E, single precision: the kernels that had it
    # (ColFft3, truncate-on-store forward, two columns per lane: 3 of its 36 loads are waited for -- all in the RAGGED-lanes
    # path, 24 guarded per-element loads that only edge tiles take and, since round 5, the one lane per input row whose two
    # columns straddle a wrapped row (ColParams::in_wrap); the 12 loads of the full-lanes path are in flight together)
    bad = [r for r in isa_scan.scan(e_s) if r[0] in ("C2RFft", "ColFft3", "ColFft3S", "R2CFft", "RowFft", "ColFft")
           and not (r[0] == "ColFft3" and r[2].endswith("ELi2EEENS_9ColParamsIfEEEEvT0_") and r[3] == 36 and r[4] <= 3)]
    assert not bad, bad


# Kernels that STILL wait for loads one at a time by the same count (profiles/r04_serialised_loads.txt): none is
# on a BASELINE path.  Each is an expected failure -- strict: the day one of them is repaired its case turns into an
# unexpected pass, i.e. a failure that says "take me off this list", so the list can only shrink visibly.  Two translation
# units are compiled for it (the double-precision groups C and K, ~100 s side by side); the 30-values plans of group M
# (c2r of 720 ... 1800 with the column limit: 22 - 46 of ~150 loads; wave-packed r2c of 600 / 900: 25 - 27 of 112) are on the
# list in the profile but not compiled here (that unit alone takes three minutes).
STILL_SERIALISED = [
    ("kernels_c_d", "C2RFft", 2048),     # c2r of real length 4096: the mirrored bin comes from memory (TPT > 64: no wave shuffle)
    ("kernels_c_d", "RowFft", 2048),     # its z-chunked c2c sibling (12 of 43)
    ("kernels_k_d", "ColFft", 2000),     # 20-values plan: cached twiddle loads of the first pass / the masked-load variants
    ("kernels_k_d", "RowFft", 2000),
]
_ASM_CACHE = {}


@pytest.fixture(scope="module")
def known_units(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = str(tmp_path_factory.mktemp("isa_known"))
    units = sorted({u for u, _, _ in STILL_SERIALISED})
    with ThreadPoolExecutor(len(units)) as ex:
        for u, path in zip(units, ex.map(lambda u: _asm(u, d), units)):
            _ASM_CACHE[u] = path
    return _ASM_CACHE


@pytest.mark.parametrize("unit,family,n", STILL_SERIALISED)
@pytest.mark.xfail(strict=True, reason="listed as still serialised in profiles/r04_serialised_loads.txt")
def test_kernels_known_to_wait_for_their_loads_one_at_a_time(known_units, unit, family, n):
    import isa_scan
    bad = [r for r in isa_scan.scan(known_units[unit]) if r[0] == family and r[1] == n]
    assert not bad, bad
