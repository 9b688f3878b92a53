"""2-D transforms on MI355X: counterpart of mpiFFT4py/line.py (class R2C, line.py:41-340).

Real data (N0/P, N1) distributed by rows, complex data (N0, Npf) distributed along ky with
the Nyquist column on the last rank (line.py:64-103).  On the device this is the x-aligned
pencil plan of the mesh (1, N0, N1) on a 1 x P grid (the `line2d` flag of the C ABI): rfft
along y, the ky-splitting all-to-all, fft along x.

Two P > 1 behaviours of the reference are handled explicitly:
  * 3/2-rule forward: the reference packs column Nf-1 of the padded y spectrum, which is not
    real there, into the imaginary part of column 0 (line.py:231) and separates the two by
    Hermitian symmetry afterwards (line.py:27-39); column 0 therefore carries Re(c0) - Im(cN)
    and the last column Re(cN).  REPRODUCED (oracle and kernels agree with the reference).
  * 2/3-rule inverse: the masked copy `fu_` and the work array `Uc_hat` are the same cached
    buffer, which is zero-filled when fetched the second time (line.py:264-266, 297;
    mpibase.py:90-93), so the reference returns zeros.  NOT reproduced: the masked transform is
    returned, as on one rank.
On one rank the reference indexes the padded rows with (fftfreq(n) * n).astype(int) (line.py:61), which
truncates e.g. 4.999999999999999 to 4 for n = 24, 28, 36, 48, ...; exact indices are used here.
"""
import numpy as np
from numpy.fft import fftfreq, rfftfreq

from . import _lib, _padding
from ._base import DistFFTBase, default_planner_effort
from .comm import as_comm
from .mpibase import datatypes, work_arrays

__all__ = ["R2C"]


class R2C(DistFFTBase):
    def __init__(self, N, L, comm, precision, padsize=1.5, threads=1, planner_effort=None):
        assert len(L) == 2
        assert len(N) == 2
        self.N = np.asarray(N, dtype=int)
        self.L = np.asarray(L).astype(float)
        self.comm = as_comm(comm)
        self.float, self.complex, self.mpitype = datatypes(precision)
        self.precision = precision
        self.num_processes = self.comm.Get_size()
        self.rank = self.comm.Get_rank()
        self.padsize = padsize
        self.threads = threads
        self.planner_effort = planner_effort if planner_effort is not None else default_planner_effort()
        self.dealias = np.zeros(0)
        self.work_arrays = work_arrays()
        self._plan = None
        self._stage = {}
        self._mask_set = False
        P = self.num_processes
        self.Np = self.N // P
        self.Nf = int(self.N[1] // 2 + 1)
        self.Npf = int(self.Np[1] // 2 + 1) if self.rank + 1 == P else int(self.Np[1] // 2)
        self.Nfp = int(padsize * self.N[1] / 2 + 1)
        self.ks = np.rint(fftfreq(int(self.N[0])) * self.N[0]).astype(int)
        N2 = self.N
        self.N = np.array([1, int(N2[0]), int(N2[1])])        # the 3-D mesh the plan sees
        try:
            self._create_plan(_lib.R2C, _lib.PENCIL_X, p1=1, line2d=True)
        finally:
            self.N = N2
        assert self._c_real_shape[1:] == tuple(self.real_shape()), (self._c_real_shape, self.real_shape())
        assert self._c_complex_shape[1:] == tuple(self.complex_shape()), (self._c_complex_shape, self.complex_shape())

    # -- shapes (line.py:76-160) ----------------------------------------------------
    def real_shape(self):
        return (int(self.Np[0]), int(self.N[1]))

    def complex_shape(self):
        return (int(self.N[0]), self.Npf)

    def global_complex_shape(self):
        return (int(self.N[0]), self.Nf)

    def global_real_shape(self):
        return (int(self.N[0]), int(self.N[1]))

    def real_local_slice(self, padsize=1):
        return (slice(int(padsize * self.rank * self.Np[0]), int(padsize * (self.rank + 1) * self.Np[0]), 1),
                slice(0, int(padsize * self.N[1])))

    def complex_local_slice(self):
        s = int(self.rank * self.Np[1] // 2)
        return (slice(0, int(self.N[0])), slice(s, s + self.Npf, 1))

    def get_N(self):
        return self.N

    def global_complex_shape_padded(self):
        return (int(self.padsize * self.N[0]), int(self.padsize * self.N[1] / 2 + 1))

    def real_shape_padded(self):
        return (int(self.padsize * self.Np[0]), int(self.padsize * self.N[1]))

    def complex_padded_xy(self):
        return (int(self.padsize * self.Np[0]), int(self.padsize * self.N[1] / 2 + 1))

    def complex_shape_padded_01(self):
        return (int(self.padsize * self.Np[0]), self.Nf)

    def complex_padded_x(self):
        return (int(self.padsize * self.N[0]), self.Npf)

    def work_shape(self, dealias):
        return self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()

    # host-side numpy helpers of the reference's API (line.py:164-175)
    def copy_to_padded_x(self, fu, fp):
        return _padding.spread(fu, fp, self.N[0], 0)

    def copy_to_padded_y(self, fu, fp):
        fp[:, :self.Nf] = fu
        return fp

    def copy_from_padded_y(self, fp, fu):
        fu[:] = fp[:, :self.Nf]
        return fu

    # -- host-side mesh helpers (line.py:105-134) -------------------------------------
    def get_local_mesh(self):
        X = np.mgrid[self.rank * self.Np[0]:(self.rank + 1) * self.Np[0], :self.N[1]].astype(self.float)
        X[0] *= self.L[0] / self.N[0]
        X[1] *= self.L[1] / self.N[1]
        return X

    def get_local_wavenumbermesh(self, scaled=True, broadcast=False, eliminate_highest_freq=False):
        kx = fftfreq(int(self.N[0]), 1. / self.N[0])
        ky = rfftfreq(int(self.N[1]), 1. / self.N[1])
        if eliminate_highest_freq:
            for i, k in enumerate((kx, ky)):
                if self.N[i] % 2 == 0:
                    k[self.N[i] // 2] = 0
        s = self.complex_local_slice()[1]
        Ks = list(np.meshgrid(kx, ky[s], indexing='ij', sparse=True))
        if scaled is True:
            Lp = 2 * np.pi / self.L
            Ks[0] = Ks[0] * Lp[0]
            Ks[1] = Ks[1] * Lp[1]
        if broadcast is True:
            return [np.broadcast_to(k, self.complex_shape()) for k in Ks]
        return Ks

    def get_dealias_filter(self):
        K = self.get_local_wavenumbermesh()
        kmax = 2. / 3. * (self.N // 2 + 1)
        return np.array((abs(K[0]) < kmax[0]) * (abs(K[1]) < kmax[1]), dtype=np.uint8)

    # -- transforms (line.py:177-338) ---------------------------------------------------
    def fft2(self, u, fu, dealias=None):
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        ushape = self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()
        assert tuple(u.shape) == ushape
        return self._run(True, u, fu, dealias, ushape, self.float, self.complex_shape(), self.complex)

    def ifft2(self, fu, u, dealias=None):
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        ushape = self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()
        assert tuple(u.shape) == ushape
        return self._run(False, fu, u, dealias, self.complex_shape(), self.complex, ushape, self.float)
