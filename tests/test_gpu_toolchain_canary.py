"""GPU: a CANARY for the two hipcc miscompile workarounds that live in the hot kernels (csrc/fft_core.h MFFT_LAUNDER_MODE,
csrc/fft_kernels.h row_thread_index).  ROCm 7.2 miscompiles the contiguous-axis kernels of the 30- and 42-values-per-thread
plans (round 6: and of the 70-values plans of 35 * 2^a) when a thread's index inside its transform has a known power-of-two
range (480, 336, 672, 560: a tenth to a third of the
bins wrong on the device, exact in the CPU emulator); the cure -- hiding the range from the twiddle index -- costs the z
stages of those plans time.  Nothing else would tell when a toolchain update FIXES the bug (the cure could go) or MOVES it
(the cure might stop working), so this test builds the reproducer tools/rowcheck2.hip twice on the GPU box:

  * mode 0 (no cure) is EXPECTED to give wrong bins on exactly the known plans -- an xfail(strict)-style assertion: the day
    every line says "ok", this test goes red with the instruction to retire the cure;
  * mode 4 (what ships) must be exact on every plan.

The reference has no counterpart (its FFTs are FFTW / pocketfft: serialFFT/pyfftw_fft.py:26-203)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from gpu_util import have_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (plan, precision) pairs whose UNCURED kernels are wrong under ROCm 7.2 (profiles/r05_miscompile_cure_modes.txt)
KNOWN_WRONG = {("480", "fp64"), ("336", "fp64"), ("672", "fp64"), ("480", "fp32"), ("672", "fp32"), ("560", "fp32")}


@pytest.fixture(scope="module")
def rowcheck():
    if not have_gpu():
        pytest.fail("no GPU visible")
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this box: the canary needs to compile its reproducer")
    tmp = tempfile.mkdtemp(prefix="mfft_canary_")
    out = {}
    try:
        procs = {}
        for mode in (0, 4):
            exe = os.path.join(tmp, "rowcheck2_%d" % mode)
            procs[mode] = (exe, subprocess.Popen(
                [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unknown-pragmas", "-Wno-unused-result",
                 "-I" + os.path.join(ROOT, "mpifft4py_amd", "csrc"), "-DMFFT_LAUNDER_MODE=%d" % mode,
                 os.path.join(ROOT, "tools", "rowcheck2.hip"), "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        for mode, (exe, p) in procs.items():
            log = p.communicate(timeout=600)[0].decode()
            assert p.returncode == 0, log[-3000:]
            r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stdout + r.stderr
            rows = {}
            for line in r.stdout.splitlines():
                m = re.match(r"mode (\d)\s+(\d+) \S+\s+(fp\d\d) .* c2c rel-L2 (\S+) (ok|WRONG)", line)
                if m:
                    rows[(m.group(2), m.group(3))] = (float(m.group(4)), m.group(5))
            assert len(rows) == 10, r.stdout
            out[mode] = rows
        yield out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_shipped_cure_is_exact(rowcheck):
    """MFFT_LAUNDER_MODE 4 (fft_core.h: the range of j hidden from the twiddle index only): every plan exact."""
    bad = {k: v for k, v in rowcheck[4].items() if v[1] != "ok"}
    assert not bad, "the shipped miscompile cure no longer works with this toolchain: %r" % bad


def test_uncured_kernels_are_still_miscompiled(rowcheck):
    """Without the cure the known plans must still come out wrong.  If this fails because everything is "ok": the toolchain
    has fixed the miscompile -- build with -DMFFT_LAUNDER_MODE=0, re-run the row / r2c / c2r stage tests and the sweep, and
    retire the cure (it costs the z stages of the 30- / 42-values plans: r2c of 360 / 600 / 720 points 0.20 / 0.31 / 0.54 ms
    cured in the cheapest place against 0.21 / 0.30 / 0.52 without; c2r keeps the dearer cure at j's origin).  If it fails
    because OTHER plans are wrong: the bug has moved, and the cure's list of plans (fft_core.h launder_plan) must follow."""
    wrong = {k for k, v in rowcheck[0].items() if v[1] != "ok"}
    assert wrong, "hipcc no longer miscompiles the uncured kernels: retire MFFT_LAUNDER_MODE (see this test's docstring)"
    assert wrong == KNOWN_WRONG, "the miscompile has moved: wrong without the cure %r, known %r" % (sorted(wrong), sorted(KNOWN_WRONG))
