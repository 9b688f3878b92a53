import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, Pencil_R2C, Slab_R2C, from_env
comm = from_env(); rank, P = comm.Get_rank(), comm.Get_size()
n = int(os.environ.get("MP_N", "256"))
N = np.array([n]*3); L = np.array([2*np.pi]*3)
A = np.random.default_rng(1).random(tuple(N))
B2 = np.fft.rfftn(A)
def rel(x, r): return float(np.linalg.norm((x-r).ravel())/np.linalg.norm(r.ravel()))
for pipeline in (1, 2, 4, 8, -2, -4, -8):
    F = Slab_R2C(N, L, comm, "double", pipeline=pipeline)
    u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
    fu = DeviceArray.empty(F.complex_shape(), F.complex); u2 = DeviceArray.empty(F.real_shape(), F.float)
    for _ in range(3):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync()
    e1, e2 = rel(fu.get(), B2[F.complex_local_slice()]), rel(u2.get(), A[F.real_local_slice()])
    assert e1 < 1e-10 and e2 < 1e-10, (pipeline, e1, e2)
if P >= 4:
    for align in "XY":
        F = Pencil_R2C(N, L, comm, "double", communication="Alltoallw", alignment=align)
        u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
        fu = DeviceArray.empty(F.complex_shape(), F.complex); u2 = DeviceArray.empty(F.real_shape(), F.float)
        F.fftn(u, fu); F.ifftn(fu, u2); F.sync()
        assert rel(fu.get(), B2[F.complex_local_slice()]) < 1e-10 and rel(u2.get(), A[F.real_local_slice()]) < 1e-10
comm.barrier()
if rank == 0: print("BIG_OK", P, n)
