import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray, _lib
N = np.array([1024]*3); L = np.array([2*np.pi]*3)
F = Slab_R2C(N, L, SelfComm(0), "double")
u = DeviceArray.random(F.real_shape(), F.float, seed=1)
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
for timing in (False, True):
    F.enable_timing(timing)
    for _ in range(3):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync(); _lib.call("mfft_device_sync")
    F.reset_timing()
    for K in (1, 2, 5, 10, 20):
        F.sync(); _lib.call("mfft_device_sync")
        t0 = time.perf_counter()
        for i in range(K):
            F.fftn(u, fu); F.ifftn(fu, u2)
        t1 = time.perf_counter()
        F.sync()
        t2 = time.perf_counter()
        _lib.call("mfft_device_sync")
        t3 = time.perf_counter()
        print("timing=%s K=%2d enqueue %.2f ms, plan sync +%.2f ms, device sync +%.2f ms -> %.2f ms/step" % (
            timing, K, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3/K))
        F.reset_timing()
