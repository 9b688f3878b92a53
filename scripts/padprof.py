"""Stage times of one 3/2-rule ifftn+fftn pair (developer tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpifft4py_amd import Slab_R2C, Pencil_R2C, SelfComm, DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "slab"      # slab | X | Y
prec = sys.argv[3] if len(sys.argv) > 3 else "double"
N = np.array([n]*3); L = np.array([2*np.pi]*3)
if kind == "slab":
    F = Slab_R2C(N, L, SelfComm(0), prec)
else:
    F = Pencil_R2C(N, L, SelfComm(0), prec, communication="Alltoallw", alignment=kind, allow_single=True)
fu = DeviceArray.random(F.complex_shape(), F.complex, seed=1)
up = DeviceArray.empty(F.real_shape_padded(), F.float)
fu2 = DeviceArray.empty(F.complex_shape(), F.complex)
F.enable_timing(True)
for _ in range(2):
    F.ifftn(fu, up, '3/2-rule'); F.fftn(up, fu2, '3/2-rule')
F.sync(); F.reset_timing()
t=time.perf_counter()
for _ in range(5):
    F.ifftn(fu, up, '3/2-rule'); F.fftn(up, fu2, '3/2-rule')
F.sync()
print("n=%d %s %s padded pair ms %.3f" % (n, kind, prec, (time.perf_counter()-t)/5*1e3))
print(" ".join("%s=%.3f" % (k, v[0]/max(v[1],1)) for k,v in sorted(F.stage_times().items())))
