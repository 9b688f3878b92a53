#!/usr/bin/env python3
"""One fresh process = one line: device addresses of the arrays of a slab R2C pair and its stage times (round 6: what IS a
"placement mode"?  the same binary lands 3 - 8 % apart from process to process, all four strided stages together).
    python scripts/placement_modes.py N [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = np.array([n, n, n])
F = Slab_R2C(N, np.array([2 * np.pi] * 3), SelfComm(0), "double")
u = DeviceArray.random(F.real_shape(), F.float, seed=1)
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
F.enable_timing(True)
for _ in range(3):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync(); F.reset_timing()
t = time.perf_counter()
for _ in range(reps):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
dt = (time.perf_counter() - t) / reps
st = {k: round(v[0] / max(v[1], 1), 3) for k, v in sorted(F.stage_times().items())}
print("%d^3 pair %.3f ms u %#x fu %#x u2 %#x (mod 2MiB: %#x %#x %#x) %s" % (n, dt * 1e3, u.ptr, fu.ptr, u2.ptr, u.ptr % (2 << 20),
      fu.ptr % (2 << 20), u2.ptr % (2 << 20), st))
