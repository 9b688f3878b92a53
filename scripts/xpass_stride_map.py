"""Which row pitches slow the strided pass down (developer tool, round 4)?  Out-of-place c2c along axis 0 of a
(1024, 1, cols) complex128 array for pitches base + delta; GB/s = 2 x bytes / best time.  python scripts/xpass_stride_map.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib
from xpass_kernel_ab import timed

_lib.load()
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
deltas = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144]
bases = [1 << 19, 3 << 18, 1 << 20, 5 << 18, 3 << 19, 1 << 21, 3 << 20, 1 << 22]
mx = (max(bases) + max(deltas)) // 16
A = DeviceArray.random((n0, 1, mx), np.complex128, seed=1)
B = DeviceArray.empty((n0, 1, mx), np.complex128)
print("rows %d; pitch = base + delta (bytes); entries: GB/s" % n0)
print("%10s " % "base" + " ".join("%7d" % d for d in deltas))
for b in bases:
    row = []
    for d in deltas:
        cols = (b + d) // 16
        s = (ctypes.c_int64 * 3)(n0, 1, cols)
        fn = lambda: _lib.call("mfft_c2c_axis", A.ptr, B.ptr, s, 0, 0, 1)
        fn()
        mn, _ = timed(fn, 5)
        row.append(2.0 * n0 * cols * 16 / mn / 1e6)
    print("%10d " % b + " ".join("%7.0f" % x for x in row), flush=True)
