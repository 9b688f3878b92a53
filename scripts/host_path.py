"""PCIe-inclusive rate of the numpy-in / numpy-out calling convention (the reference's own):
python scripts/host_path.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = np.array([n] * 3)
F = Slab_R2C(N, np.array([2 * np.pi] * 3), SelfComm(0), "double")
u = np.random.default_rng(0).random(tuple(N))
fu = np.zeros(F.complex_shape(), dtype=complex)
u2 = np.zeros(F.real_shape())
for it in range(4):
    t0 = time.perf_counter()
    fu = F.fftn(u, fu)
    t1 = time.perf_counter()
    u2 = F.ifftn(fu, u2)
    t2 = time.perf_counter()
    print("n=%d iter %d: fftn %.1f ms, ifftn %.1f ms  (%.1f GB/s effective over in+out bytes)"
          % (n, it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, 2 * (u.nbytes + fu.nbytes) / (t2 - t0) / 1e9))
print("roundtrip err", float(np.abs(u2 - u).max()))
