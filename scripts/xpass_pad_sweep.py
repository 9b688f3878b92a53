"""Out-of-place x pass of (n0, cols) rows as a function of the row pitch cols + k lines of 128 B (developer tool, round 4):
which pitches are slow?  python scripts/xpass_pad_sweep.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib
from xpass_kernel_ab import timed

_lib.load()


def sweep(label, n0, cols, dtype, pads=(0, 1, 2, 3, 5)):
    es = np.dtype(dtype).itemsize
    prec = _lib.precision_code(dtype)
    line = 128 // es
    mx = cols + max(pads) * line
    A = DeviceArray.random((n0, 1, mx), dtype, seed=1)
    B = DeviceArray.empty((n0, 1, mx), dtype)
    out = []
    for k in pads:
        w = cols + k * line
        s = (ctypes.c_int64 * 3)(n0, 1, w)
        fn = lambda: _lib.call("mfft_c2c_axis", A.ptr, B.ptr, s, 0, 0, prec)
        fn()
        mn, med = timed(fn, 5)
        out.append("+%d: %.3f" % (k, mn))
    s = (ctypes.c_int64 * 3)(n0, 1, cols)
    fn = lambda: _lib.call("mfft_c2c_axis", A.ptr, A.ptr, s, 0, 0, prec)
    fn()
    mn, med = timed(fn, 5)
    print("%-52s pitch %9d B (%% 65536 = %5d, %% 4096 = %4d): in place %.3f ms; out of place by lines added: %s"
          % (label, cols * es, (cols * es) % 65536, (cols * es) % 4096, mn, "  ".join(out)), flush=True)
    A.free(); B.free()


if __name__ == "__main__":
    sweep("HEADLINE 1024^3 fp64 R2C, one rank (1024 x 1024*513)", 1024, 1024 * 513, np.complex128)
    sweep("512^3 fp64 R2C, one rank (512 x 512*257)", 512, 512 * 257, np.complex128)
    sweep("1024^3 fp32 R2C, one rank (1024 x 1024*513)", 1024, 1024 * 513, np.complex64)
    for P in (2, 4, 8):
        sweep("1024^3 fp64 R2C slab, %d ranks" % P, 1024, 1024 // P * 513, np.complex128)
    sweep("1024^3 fp64 R2C pencil X 4x2 (1024 x 256*256)", 1024, 256 * 256, np.complex128)
    sweep("1024^3 fp64 R2C pencil X 4x2 (1024 x 256*257)", 1024, 256 * 257, np.complex128)
    sweep("1024^3 fp64 R2C pencil Y 4x2 inv (1024 x 512*128)", 1024, 512 * 128, np.complex128)
    sweep("1024^3 fp64 R2C pencil Y 4x2 inv (1024 x 512*129)", 1024, 512 * 129, np.complex128)
    sweep("768^3 fp64 R2C one rank (768 x 768*385)", 768, 768 * 385, np.complex128)
    sweep("1536^3 fp64 R2C one rank (1536 x 1536*769)", 1536, 1536 * 769, np.complex128, pads=(0, 1, 2))
