"""Workload for profiling the data-movement kernels (box_copy*, mask_kernel) that the fused paths normally avoid
(developer tool; run under rocprofv3):  the copy-based z-chunk pack / unpack of the pencils (MFFT_NO_ZFUSE), the
copy-based 3/2-rule pad / truncate (MFFT_NO_PAD_FUSION) and the 2/3-rule mask, 512^3 fp64."""
import os, sys
import numpy as np
os.environ["MFFT_NO_ZFUSE"] = "1"
os.environ["MFFT_NO_PAD_FUSION"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Pencil_R2C, SelfComm, Slab_R2C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)


def pencils(comm):
    for align in ("X", "Y"):
        F = Pencil_R2C(N, L, comm, "double", communication="Alltoallw", alignment=align, pipeline=1)
        u = DeviceArray.random(F.real_shape(), F.float, seed=3 + comm.Get_rank())
        fu = DeviceArray.empty(F.complex_shape(), F.complex)
        for _ in range(3):
            F.fftn(u, fu); F.ifftn(fu, u)
        F.sync(); comm.barrier()


g = LocalGroup(8, devices=[0] * 8)
g.run(pencils)
g.free()
F = Slab_R2C(N, L, SelfComm(0), "double")
fu = DeviceArray.random(F.complex_shape(), F.complex, seed=5)
up = DeviceArray.empty(F.real_shape_padded(), F.float)
u = DeviceArray.empty(F.real_shape(), F.float)
for _ in range(3):
    F.ifftn(fu, up, dealias="3/2-rule"); F.fftn(up, fu, dealias="3/2-rule")       # box_copy pad / truncate (+ fold)
    F.ifftn(fu, u, dealias="2/3-rule")                                             # mask_kernel
F.sync()
# the slab pack / unpack of the reference as standalone kernels at a size that fills the GPU: the (128, 1024, 513)
# complex128 slab of one of 8 ranks at 1024^3 (1.08 GB each way); slab.py:403, cython/maths.pyx:21-31
from mpifft4py_amd import _lib
P, Np0, Np1, Nf = 8, 128, 128, 513
T = DeviceArray.random((Np0, P * Np1, Nf), np.complex128, seed=9)
M = DeviceArray.empty((P, Np0, Np1, Nf), np.complex128)
for _ in range(3):
    _lib.call("mfft_slab_pack", T.ptr, M.ptr, P, Np0, Np1, Nf, _lib.precision_code("double"))
    _lib.call("mfft_slab_unpack", M.ptr, T.ptr, P, Np0, Np1, Nf, _lib.precision_code("double"))
# z-chunk pack of a pencil rank at 1024^3 (4 x 2 grid: (256, 512, 513) rows cut into two chunks) as ONE rank's launch
Z = DeviceArray.random((256 * 512, 513), np.complex128, seed=10)
S = DeviceArray.empty((256 * 512 * 513,), np.complex128)
lib = _lib.load()
print("PACK_WORKLOAD_OK")
