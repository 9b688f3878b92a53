"""Does the slow mode of the inverse x pass / c2r (run-to-run spread) follow the WORK buffer?
One process, the same user arrays, K plans created one after the other and all kept alive, so that every plan's work
buffer lies in other physical memory; stage times of each.  python scripts/reroll_probe.py [K]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = np.array([1024] * 3); L = np.array([2 * np.pi] * 3)
keep = []
u = DeviceArray.random((1024, 1024, 1024), np.float64, seed=1)
fu = DeviceArray.empty((1024, 1024, 513), np.complex128)
u2 = DeviceArray.empty((1024, 1024, 1024), np.float64)
for k in range(K):
    F = Slab_R2C(N, L, SelfComm(0), "double")
    keep.append(F)
    for _ in range(3):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync()
    F.enable_timing(True)
    for _ in range(8):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync()
    st = F.stage_times()
    print("plan %d: " % k + " ".join("%s=%.3f" % (n, v[0] / max(v[1], 1)) for n, v in sorted(st.items())), flush=True)
    F.enable_timing(False)
