"""The per-rank passes of the BASELINE multi-GPU configs, each alone on the device (developer tool, round 4): which of the
kernels a rank runs at 1024^3 over 8 ranks (and config 5) is far from what the one-rank transform reaches?  Stage-level entry
points (mfft_c2c_axis / mfft_r2c_last / mfft_c2r_last) on contiguous arrays of the rank-local shapes; GB/s = algorithmic
bytes / best of 7.  python scripts/rank_shapes.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib
from xpass_kernel_ab import timed

_lib.load()


def c2c(label, shape, axis, dtype, inplace=False):
    A = DeviceArray.random(shape, dtype, seed=1)
    B = A if inplace else DeviceArray.empty(shape, dtype)
    s = (ctypes.c_int64 * 3)(*shape)
    prec = _lib.precision_code(dtype)
    fn = lambda: _lib.call("mfft_c2c_axis", A.ptr, B.ptr, s, axis, 0, prec)
    fn()
    mn, _ = timed(fn)
    gb = 2.0 * A.nbytes / 1e9
    print("%-64s %8.3f ms  %5.0f GB/s" % (label, mn, gb / mn * 1e3), flush=True)
    A.free()
    if B is not A:
        B.free()


def real(label, rows, n, dtype):
    rs = (ctypes.c_int64 * 3)(1, rows, n)
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    R = DeviceArray.random((1, rows, n), dtype, seed=2)
    C = DeviceArray.empty((1, rows, n // 2 + 1), cdt)
    prec = _lib.precision_code(dtype)
    for nm, fn in (("r2c", lambda: _lib.call("mfft_r2c_last", R.ptr, C.ptr, rs, prec)),
                   ("c2r", lambda: _lib.call("mfft_c2r_last", C.ptr, R.ptr, rs, prec))):
        fn()
        mn, _ = timed(fn)
        gb = (R.nbytes + C.nbytes) / 1e9
        print("%-64s %8.3f ms  %5.0f GB/s" % (label + " " + nm, mn, gb / mn * 1e3), flush=True)
    R.free(); C.free()


if __name__ == "__main__":
    f64, c128, c64 = np.float64, np.complex128, np.complex64
    print("# one rank, 1024^3 fp64 (reference)")
    real("z: 1024*1024 rows of 1024", 1024 * 1024, 1024, f64)
    c2c("y: (1024, 1024, 513) axis 1, in place", (1024, 1024, 513), 1, c128, True)
    c2c("x: (1024, 1024*513) axis 0, in place", (1024, 1, 1024 * 513), 0, c128, True)
    print("# slab over 8 ranks")
    real("z: 128*1024 rows of 1024", 128 * 1024, 1024, f64)
    c2c("y: (128, 1024, 513) axis 1", (128, 1024, 513), 1, c128)
    c2c("y of one kz slice: (128, 1024, 128) axis 1", (128, 1024, 128), 1, c128)
    c2c("x: (1024, 128*513) axis 0", (1024, 1, 128 * 513), 0, c128)
    c2c("x: (1024, 128*513 + 8) axis 0 (padded pitch)", (1024, 1, 128 * 513 + 8), 0, c128)
    print("# pencil 4 x 2 over 8 ranks (x-aligned)")
    real("z: 256*512 rows of 1024", 256 * 512, 1024, f64)
    c2c("y: (256, 1024, 256) axis 1", (256, 1024, 256), 1, c128)
    c2c("y: (256, 1024, 257) axis 1", (256, 1024, 257), 1, c128)
    c2c("x: (1024, 256*257) axis 0", (1024, 1, 256 * 257), 0, c128)
    c2c("x: (1024, 256*257 + 8) axis 0 (padded pitch)", (1024, 1, 256 * 257 + 8), 0, c128)
    print("# pencil 4 x 2 (y-aligned)")
    c2c("x: (1024, 512*128) axis 0, in place", (1024, 1, 512 * 128), 0, c128, True)
    c2c("x: (1024, 512*129) axis 0, in place", (1024, 1, 512 * 129), 0, c128, True)
    c2c("y: (512, 1024, 128) axis 1", (512, 1024, 128), 1, c128)
    c2c("y: (512, 1024, 129) axis 1", (512, 1024, 129), 1, c128)
    print("# config 5: 2048^3 c64 pencil 4 x 2 (x-aligned), per rank")
    c2c("z: (512*1024 rows of 2048) axis 2", (512, 1024, 2048), 2, c64)
    c2c("y: (512, 2048, 1024) axis 1", (512, 2048, 1024), 1, c64)
    c2c("y: (512, 2048, 1024 + 16) axis 1 (rows one line further apart)", (512, 2048, 1040), 1, c64)
    c2c("x: (2048, 512*1024) axis 0", (2048, 1, 512 * 1024), 0, c64)
    c2c("x: (2048, 512*1024 + 16) axis 0 (padded pitch)", (2048, 1, 512 * 1024 + 16), 0, c64)
