#!/usr/bin/env python3
"""Times the fused nonlinear z stage (mfft_nlz_rows, csrc/fft_nlz.h) on its own: rows of real length n with `valid` bins,
in place on the first field as the plan runs it.  Prints ms per launch and TB/s of algorithmic traffic (9 rows of
valid bins per (x, y) row).  (the experiment builds behind profiles/r06_nlz_variants.txt selected their variants with MFFT_NLZ_VARIANT)

    python scripts/nlz_bench.py [n valid nrows precision] ..."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib  # noqa: E402

CASES = [(768, 257, 768 * 96, "double"), (1536, 513, 1536 * 24, "double"), (512, 257, 512 * 128, "double"),
         (1024, 513, 1024 * 48, "double"), (768, 257, 768 * 96, "single"), (1536, 513, 1536 * 48, "single")]


def bench(n, valid, nrows, prec, reps=10):
    cd = np.complex128 if prec == "double" else np.complex64
    line = 128 // np.dtype(cd).itemsize
    pitch = (valid + line - 1) // line * line
    a = DeviceArray.random((3, nrows, pitch), cd, seed=1)
    b = DeviceArray.random((3, nrows, pitch), cd, seed=2)
    code = _lib.precision_code(prec)
    _lib.call("mfft_nlz_rows", a.ptr, b.ptr, a.ptr, nrows, n, pitch, valid, code, 1)
    t = ctypes.c_void_p()
    _lib.call("mfft_timer_create", ctypes.byref(t))
    _lib.call("mfft_timer_start", t)
    for _ in range(reps):
        _lib.call("mfft_nlz_rows", a.ptr, b.ptr, a.ptr, nrows, n, pitch, valid, code, 0)
    ms = ctypes.c_float(0)
    _lib.call("mfft_timer_stop", t, ctypes.byref(ms))
    _lib.call("mfft_timer_destroy", t)
    ms = ms.value / reps
    gb = 9.0 * nrows * valid * np.dtype(cd).itemsize / 1e9
    print("nlz n=%-5d valid=%-4d rows=%-7d %-6s variant=%s  %8.3f ms  %6.2f TB/s algorithmic  (%.1f ns per row)"
          % (n, valid, nrows, prec, os.environ.get("MFFT_NLZ_VARIANT", "0"), ms, gb / ms, 1e6 * ms / nrows))


if __name__ == "__main__":
    args = sys.argv[1:]
    cases = CASES if not args else [(int(args[i]), int(args[i + 1]), int(args[i + 2]), args[i + 3]) for i in range(0, len(args), 4)]
    for c in cases:
        bench(*c)
