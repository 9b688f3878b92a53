"""The x pass behind an exchange as ONE rank of a multi-GPU run would execute it -- alone on the device -- in its three
forms (developer tool; round 4):
  (a) in place on a compact (N0, cols) array                      rounds 1 - 3
  (b) out of place, compact -> compact                            MFFT_NO_XPAD=1
  (c) out of place, rows one cache line further apart -> the same pitch   (the read side of plan.hip xplane_pad; the store
      side of the real pass is compact, which costs nothing: profiles/r02_power_of_two_stride.txt)
through the stage-level entry point mfft_c2c_axis (serialFFT.fft's seam), timed with HIP events.
python scripts/xpass_kernel_ab.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib

_lib.load()


def timed(fn, reps=7):
    t = ctypes.c_void_p()
    _lib.call("mfft_timer_create", ctypes.byref(t))
    out = []
    for _ in range(reps):
        ms = ctypes.c_float(0)
        _lib.call("mfft_timer_start", t)
        fn()
        _lib.call("mfft_timer_stop", t, ctypes.byref(ms))
        out.append(ms.value)
    _lib.call("mfft_timer_destroy", t)
    out.sort()
    return out[0], out[len(out) // 2]


def run(label, n0, cols, dtype):
    es = np.dtype(dtype).itemsize
    prec = _lib.precision_code(dtype)
    line = 128 // es

    def c2c(a, b, shape):
        s = (ctypes.c_int64 * 3)(*shape)
        _lib.call("mfft_c2c_axis", a.ptr, b.ptr, s, 0, 0, prec)
    A = DeviceArray.random((n0, 1, cols + line), dtype, seed=1)
    B = DeviceArray.empty((n0, 1, cols + line), dtype)
    gb = 2.0 * n0 * cols * es / 1e9
    res = []
    for name, fn in (("in place, compact", lambda: c2c(A, A, (n0, 1, cols))),
                     ("out of place, compact", lambda: c2c(A, B, (n0, 1, cols))),
                     ("out of place, +1 line", lambda: c2c(A, B, (n0, 1, cols + line)))):
        fn()
        mn, med = timed(fn)
        res.append("%s %.3f ms (%.0f GB/s)" % (name, mn, gb / mn * 1e3))
    print("%-46s rows %8d B apart: %s" % (label, cols * es, "; ".join(res)), flush=True)
    A.free(); B.free()


if __name__ == "__main__":
    run("config 5, pencil X, 8 ranks (2048 x 512*1024 c64)", 2048, 512 * 1024, np.complex64)
    run("2048^3 c64 slab, 8 ranks (2048 x 256*2048)", 2048, 256 * 2048, np.complex64)
    run("1024^3 c64 pencil X, 8 ranks (1024 x 256*512)", 1024, 256 * 512, np.complex64)
    run("1024^3 c128 slab, 2 ranks (1024 x 512*1024)", 1024, 512 * 1024, np.complex128)
    run("1024^3 c128 slab, 8 ranks (1024 x 128*1024)", 1024, 128 * 1024, np.complex128)
    run("1024^3 fp64 R2C slab, 8 ranks (1024 x 128*513)", 1024, 128 * 513, np.complex128)
    run("1024^3 fp64 R2C slab, 2 ranks (1024 x 512*513)", 1024, 512 * 513, np.complex128)
    run("1024^3 fp64 R2C pencil X, 8 ranks (1024 x 256*257)", 1024, 256 * 257, np.complex128)
    run("1024^3 fp64 R2C pencil Y inv, 8 ranks (1024 x 512*129)", 1024, 512 * 129, np.complex128)
