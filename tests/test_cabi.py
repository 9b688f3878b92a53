"""CPU: the C-ABI library builds, loads, and exports exactly the symbols that
include/mpifft4py_amd.h declares (no compute calls: there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mpifft4py_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mpifft4py_amd", "csrc"), "-j8"])
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mpifft4py_amd.h")).read()
    return sorted(set(re.findall(r"MFFT_API\s+[\w\s\*]+?\b(mfft_\w+)\s*\(", txt)))


def test_header_matches_binding(lib):
    from mpifft4py_amd import _lib
    hs = header_symbols()
    assert len(hs) >= 45
    assert hs == _lib.exported_symbols()


def test_every_symbol_is_exported(lib):
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_nm_exports_only_the_abi(lib):
    from mpifft4py_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("mfft_"))
    assert exported == header_symbols()


def test_version_and_lengths(lib):
    assert lib.mfft_version() >= 100
    for n in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 48, 96, 192, 768, 1536, 3072, 80, 640, 250, 1000, 2000):
        assert lib.mfft_length_supported(n, 0) == 1, n
    for n in (30, 240, 480, 720, 900, 960, 1200, 1440, 1800, 750, 1500, 1920, 2400, 3000, 3840):   # round 3: 3 and 5 among the factors
        assert lib.mfft_length_supported(n, 0) == 1, n
        assert lib.mfft_length_supported(2 * n, 1) == 1, n
    for n in (14, 28, 56, 112, 224, 448, 896, 1792, 3584, 8192):      # round 4: 7 * 2^a radix plans, 8192
        assert lib.mfft_length_supported(n, 0) == 1, n
        assert lib.mfft_length_supported(2 * n, 1) == 1, n
    for n in (8, 64, 1024, 2048, 8192, 16384, 48, 96, 1536, 4000):
        assert lib.mfft_length_supported(n, 1) == 1, n
    # chirp-z range: 2n-1 <= 8192, i.e. EVERY length up to 4096 (numpy_fft.py:25-46: the reference takes any n)
    for n in (7, 11, 13, 17, 36, 1001, 2047, 2049, 2100, 2688, 3600, 4093, 4095):
        assert lib.mfft_length_supported(n, 0) == 1, n
        assert lib.mfft_length_supported(n, 1) == 1, n
    assert all(lib.mfft_length_supported(n, 0) == 1 for n in range(1, 4097))
    assert all(lib.mfft_length_supported(n, 1) == 1 for n in range(2, 4097))
    # round 5: radix plans between 4096 and 8192, and EVERY other length up to 2^20 through the scratch-buffer fallback
    # (csrc/bigfft.hip) -- numpy_fft.py:25-46 takes any n
    for n in (4608, 5120, 6144, 7168, 42, 84, 168, 336, 672, 1344, 2688):
        assert lib.mfft_length_route(n, 0) == 1 and lib.mfft_length_route(2 * n, 1) == 1, n
    assert all(lib.mfft_length_supported(n, 0) == 1 for n in range(1, 8193))
    assert all(lib.mfft_length_supported(n, 1) == 1 for n in range(2, 16385))
    for n in (4097, 5000, 8191, 10007, 65536, 100000, 1 << 20):
        assert lib.mfft_length_supported(n, 0) == 1 and lib.mfft_length_supported(n, 1) == 1, n
        assert lib.mfft_length_route(n, 0) == 3, n
    assert lib.mfft_length_route(4098, 1) == 2 and lib.mfft_length_route(8190, 1) == 2      # even real rows: chirp-z of n/2 complex values
    assert lib.mfft_length_route(4099, 1) == 3 and lib.mfft_length_route(8194, 1) == 3
    assert lib.mfft_length_route(1024, 0) == 1 and lib.mfft_length_route(1001, 0) == 2
    assert lib.mfft_length_supported((1 << 20) + 1, 0) == 0 and lib.mfft_length_route(0, 0) == 0
    # per precision: 35 * 2^a has radix plans in single precision only (plans.h group S); the 3/2-rule images of round 6 in both
    assert lib.mfft_length_route_precision(1120, 0, 1) == 2 and lib.mfft_length_route_precision(1120, 0, 0) == 1
    for n in (432, 864, 1728, 1080, 1296, 1008, 2700):
        assert lib.mfft_length_route_precision(n, 0, 0) == 1 and lib.mfft_length_route_precision(n, 0, 1) == 1, n


def test_fails_loudly_without_gpu(lib):
    """On a box without a GPU the product path raises; it never falls back."""
    from mpifft4py_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    from mpifft4py_amd import Slab_R2C
    with pytest.raises(_lib.MfftError):
        Slab_R2C(np.array([8, 8, 8]), np.array([1., 1., 1.]), None, "double")


def test_emulator_passes():
    """The fibre-based workgroup emulator runs the exact kernel bodies on the CPU
    against a long-double DFT for every plan in plans.h."""
    import glob
    csrc = os.path.join(ROOT, "mpifft4py_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-j", "7", "emu"])
    parts = sorted(glob.glob(os.path.join(csrc, "build", "emu_test_[0-9]")) + glob.glob(os.path.join(csrc, "build", "emu_test_[0-9][0-9]")))
    assert len(parts) == 18, parts         # the plan list is split over eighteen binaries (Makefile: EMU_PARTS; 12 = the fused nonlinear z stage, 13 = radix 70, 14 - 17 = the 3/2-rule image plans of round 6)
    procs = [subprocess.Popen([p], stdout=subprocess.PIPE) for p in parts]
    for p, proc in zip(parts, procs):
        out = proc.communicate()[0].decode()
        assert proc.returncode == 0 and "EMU TESTS PASSED" in out, (p, out[-2000:])
