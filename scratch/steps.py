import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray, _lib
N = np.array([1024]*3); L = np.array([2*np.pi]*3)
t0 = time.perf_counter()
F = Slab_R2C(N, L, SelfComm(0), "double")
u = DeviceArray.random(F.real_shape(), F.float, seed=1)
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
_lib.call("mfft_device_sync")
print("setup %.3f s" % (time.perf_counter()-t0))
ts = []
for i in range(25):
    t = time.perf_counter()
    F.fftn(u, fu); F.ifftn(fu, u2); F.sync()
    ts.append((time.perf_counter()-t)*1e3)
print(" ".join("%.1f" % x for x in ts))
