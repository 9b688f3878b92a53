"""Stage times of a slab R2C pair on one GPU (developer tool): python scripts/r2cprof.py n precision [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "double"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)
F = Slab_R2C(N, L, SelfComm(0), prec)
u = DeviceArray.random(F.real_shape(), F.float, seed=1)
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
for _ in range(3):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
t = time.perf_counter()
for _ in range(reps):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
dt = (time.perf_counter() - t) / reps
F.enable_timing(True)
for _ in range(reps):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
print("n=%d %s R2C pair %.3f ms" % (n, prec, dt * 1e3))
print(" ".join("%s=%.3f" % (k, v[0] / max(v[1], 1)) for k, v in sorted(F.stage_times().items())))
