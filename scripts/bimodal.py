"""Does the inverse x pass (out-of-place, fu -> work buffer) depend on where the work buffer lands?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
N = np.array([1024] * 3); L = np.array([2 * np.pi] * 3)
comm = SelfComm(0)
plans = []
u = DeviceArray.random((1024, 1024, 1024), np.float64, seed=1)
fu = DeviceArray.empty((1024, 1024, 513), np.complex128)
u2 = DeviceArray.empty((1024, 1024, 1024), np.float64)
for i in range(6):
    F = Slab_R2C(N, L, comm, "double")
    F.enable_timing(True)
    for _ in range(2):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync(); F.reset_timing()
    for _ in range(6):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync()
    st = F.stage_times()
    print("plan %d: " % i + " ".join("%s=%.3f" % (k, v[0] / max(v[1], 1)) for k, v in sorted(st.items())))
    plans.append(F)     # keep the work buffers alive so that the next plan gets different addresses
