"""Stage times of a slab C2C pair (developer tool): python scripts/c2cprof.py n precision"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_C2C, SelfComm, DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "single"
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)
F = Slab_C2C(N, L, SelfComm(0), prec)
u = DeviceArray.random(F.original_shape(), F.complex, seed=1) if hasattr(DeviceArray, "random") else None
fu = DeviceArray.empty(F.transformed_shape(), F.complex)
u2 = DeviceArray.empty(F.original_shape(), F.complex)
F.enable_timing(True)
for _ in range(2):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync(); F.reset_timing()
t = time.perf_counter()
for _ in range(5):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
dt = (time.perf_counter() - t) / 5
V = u.nbytes
a, b = u.leading(0, 1).get(), u2.leading(0, 1).get()
rt = float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(a.ravel()))
print("n=%d %s C2C pair %.3f ms  (12 V / t = %.0f GB/s = %.1f%% of 8 TB/s)  round trip %.1e%s"
      % (n, prec, dt * 1e3, 12 * V / dt / 1e9, 12 * V / dt / 8e12 * 100, rt, "" if rt < (1e-5 if prec == "single" else 1e-12) else "  WRONG"))
print(" ".join("%s=%.3f" % (k, v[0] / max(v[1], 1)) for k, v in sorted(F.stage_times().items())))
