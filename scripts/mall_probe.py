"""Is a strided-axis pass faster when its data sits in the Infinity Cache?  One kz-tile of a 1024^3 transform is a
(1024, 1024, 8) c128 array = 134 MB; run y and x passes on it repeatedly and compare with the 8.6 GB array."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib
_lib.load()

def run(shape, reps):
    a = DeviceArray.empty(shape, np.complex128)
    _lib.call("mfft_memset", a.ptr, 0, a.nbytes)
    s = (ctypes.c_int64 * 3)(*shape)
    for ax in (1, 0):
        _lib.call("mfft_c2c_axis", a.ptr, a.ptr, s, ax, 0, 1)
    _lib.call("mfft_device_sync")
    out = {}
    for ax, name in ((1, "y"), (0, "x")):
        t = time.perf_counter()
        for _ in range(reps):
            _lib.call("mfft_c2c_axis", a.ptr, a.ptr, s, ax, 0, 1)
        _lib.call("mfft_device_sync")
        dt = (time.perf_counter() - t) / reps
        out[name] = (dt * 1e3, 2 * a.nbytes / dt / 1e9)
    t = time.perf_counter()
    for _ in range(reps):
        _lib.call("mfft_c2c_axis", a.ptr, a.ptr, s, 1, 0, 1)
        _lib.call("mfft_c2c_axis", a.ptr, a.ptr, s, 0, 0, 1)
    _lib.call("mfft_device_sync")
    dt = (time.perf_counter() - t) / reps
    out["y+x"] = (dt * 1e3, 4 * a.nbytes / dt / 1e9)
    return a.nbytes, out

for shape, reps in (((1024, 1024, 8), 50), ((1024, 1024, 4), 50), ((1024, 1024, 16), 30), ((1024, 1024, 64), 10), ((1024, 1024, 512), 3)):
    nb, o = run(shape, reps)
    print("%-18s %7.1f MB  " % (shape, nb / 1e6) + "  ".join("%s: %.3f ms %.0f GB/s" % (k, v[0], v[1]) for k, v in o.items()))
