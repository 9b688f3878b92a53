"""mpifft4py_amd -- MI355X-native distributed 3-D FFT with the mpiFFT4py API.

Same public names as mpiFFT4py/__init__.py:1-8:

    from mpifft4py_amd import Slab_R2C, Pencil_R2C, work_arrays, datatypes, ...
    FFT = Slab_R2C(N, L, comm, "double")
    fu = FFT.fftn(u, fu);  u = FFT.ifftn(fu, u)

`comm` is None / SelfComm() for one GPU, comm.from_env() for one process per
GPU (RCCL), a LocalGroup rank for single-process multi-rank, or an mpi4py
communicator.  Arrays may be numpy (copied) or DeviceArray (HBM-resident).
"""
from numpy.fft import fftfreq, rfftfreq  # noqa: F401

from .serialFFT import *  # noqa: F401,F403
from .slab import R2C as Slab_R2C  # noqa: F401
from .slab import C2C as Slab_C2C  # noqa: F401
from .pencil import R2C as Pencil_R2C  # noqa: F401
from .pencil import C2C as Pencil_C2C  # noqa: F401  (extension: no reference counterpart)
from .line import R2C as Line_R2C  # noqa: F401
from .mpibase import work_arrays, datatypes, empty, zeros  # noqa: F401
from .device import DeviceArray  # noqa: F401
from .comm import SelfComm, LocalGroup, LayoutComm, DistComm, from_env, from_mpi4py  # noqa: F401

__version__ = "0.1.0"
