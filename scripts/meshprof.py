"""Stage times of a slab R2C pair on an arbitrary mesh (developer tool): python scripts/meshprof.py N0 N1 N2 [precision [3/2-rule]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
N = np.array([int(x) for x in sys.argv[1:4]])
prec = sys.argv[4] if len(sys.argv) > 4 else "double"
F = Slab_R2C(N, np.array([2 * np.pi] * 3), SelfComm(0), prec)
u = DeviceArray.random(F.real_shape(), F.float, seed=1)
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
F.enable_timing(True)
for _ in range(2):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync(); F.reset_timing()
t = time.perf_counter()
for _ in range(5):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
dt = (time.perf_counter() - t) / 5
R, C = u.nbytes, fu.nbytes
st = F.stage_times()
print("mesh %s %s pair %.3f ms = %.0f GB/s (%.1f%% of 8 TB/s)" % (list(N), prec, dt * 1e3, 2 * (R + 5 * C) / dt / 1e9, 2 * (R + 5 * C) / dt / 8e10))
for k, v in sorted(st.items()):
    ms = v[0] / max(v[1], 1)
    b = (R + C) if k.endswith("_z") else 2 * C
    print("  %-6s %.3f ms  %.0f GB/s" % (k, ms, b / ms / 1e6))
if len(sys.argv) > 5 and sys.argv[5] == "3/2-rule":
    del u2
    up = DeviceArray.empty(F.real_shape_padded(), F.float)
    for _ in range(2):
        F.ifftn(fu, up, "3/2-rule"); F.fftn(up, fu, "3/2-rule")
    F.sync(); F.reset_timing()
    t = time.perf_counter()
    for _ in range(5):
        F.ifftn(fu, up, "3/2-rule"); F.fftn(up, fu, "3/2-rule")
    F.sync()
    print("mesh %s %s 3/2-rule pair %.3f ms" % (list(N), prec, (time.perf_counter() - t) / 5 * 1e3))
    for k, v in sorted(F.stage_times().items()):
        if v[1]:
            print("  %-6s %.3f ms (3/2-rule)" % (k, v[0] / v[1]))
