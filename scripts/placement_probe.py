"""Does the time of the out-of-place inverse x pass (fu -> W) and of the c2r pass (W -> u) depend on where W lies
relative to fu / u?  (the pair is bimodal from process to process.)  One process, W carved out of
a larger allocation at different offsets, each offset timed 5 times.  python scripts/placement_probe.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib
_lib.load()
N, NF = 1024, 513
cbytes = N * N * NF * 16
fu = DeviceArray.random((N, N, NF), np.complex128, seed=1)
u = DeviceArray.empty((N, N, N), np.float64)
pool = DeviceArray.empty(((cbytes + (64 << 20)) // 16,), np.complex128)        # W + 64 MiB of slack
shape_c = (ctypes.c_int64 * 3)(N, N, NF)
shape_r = (ctypes.c_int64 * 3)(N, N, N)
print("fu at %#x (mod 2 MiB %#x, mod 1 GiB %#x), u at %#x, pool at %#x" % (fu.ptr, fu.ptr % (2 << 20), fu.ptr % (1 << 30), u.ptr, pool.ptr))


def timed(fn, reps=5):
    fn(); _lib.call("mfft_device_sync")
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    _lib.call("mfft_device_sync")
    return (time.perf_counter() - t) / reps * 1e3


print("%12s %14s %14s | %9s %9s" % ("offset", "W mod 2 MiB", "(W-fu) mod 1GiB", "x ms", "c2r ms"))
for off in [0, 128 << 10, 256 << 10, 384 << 10, 1 << 20, (1 << 20) + (128 << 10), 2 << 20, 3 << 20, (3 << 20) + (640 << 10), 8 << 20,
            (16 << 20) + (128 << 10), 32 << 20, (48 << 20) + (896 << 10), 63 << 20]:
    W = pool.ptr + off
    tx = timed(lambda: _lib.call("mfft_c2c_axis", fu.ptr, W, shape_c, 0, 1, 1))
    tz = timed(lambda: _lib.call("mfft_c2r_last", W, u.ptr, shape_r, 1))
    print("%12d %#14x %#14x | %9.3f %9.3f" % (off, W % (2 << 20), (W - fu.ptr) % (1 << 30), tx, tz))
