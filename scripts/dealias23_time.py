"""Time of one '2/3-rule' ifftn beside the plain one (developer tool): python3 scripts/dealias23_time.py 1200 double"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "double"
F = Slab_R2C(np.array([n] * 3), np.array([2 * np.pi] * 3), SelfComm(0), prec)
fu = DeviceArray.random(F.complex_shape(), F.complex, seed=1)
u = DeviceArray.empty(F.real_shape(), F.float)
for mode in (None, '2/3-rule'):
    for _ in range(2):
        F.ifftn(fu, u, mode)
    F.sync()
    t = time.perf_counter()
    for _ in range(5):
        F.ifftn(fu, u, mode)
    F.sync()
    print("n=%d %s ifftn(%s) ms %.3f" % (n, prec, mode, (time.perf_counter() - t) / 5 * 1e3))
