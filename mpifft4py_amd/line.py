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

from . import _lib, _padding
from ._base import DistFFTBase, default_planner_effort
from ._mesh import dft_modes

__all__ = ["R2C"]


class R2C(DistFFTBase):
    def __init__(self, N, L, comm, precision, padsize=1.5, threads=1, planner_effort=None):
        assert len(L) == 2
        assert len(N) == 2
        self._init_common(N, L, comm, precision, None, padsize, threads,
                          planner_effort if planner_effort is not None else default_planner_effort(), ndim=2)
        self.L = np.asarray(L)
        P = self.num_processes
        self.Np = self.N // P
        self.Nf = int(self.N[1] // 2 + 1)
        self.Npf = int(self.Np[1] // 2 + 1) if self.rank + 1 == P else int(self.Np[1] // 2)
        self.Nfp = int(padsize * self.N[1] / 2 + 1)
        self.ks = dft_modes(self.N[0])
        # the 3-D mesh the plan sees: (1, Nx, Ny) on a 1 x P grid
        self._describe(_lib.R2C, _lib.PENCIL_X, mesh=(1, int(self.N[0]), int(self.N[1])), p1=1, line2d=True)
        assert self._c_real_shape[1:] == tuple(self.real_shape()), (self._c_real_shape, self.real_shape())
        assert self._c_complex_shape[1:] == tuple(self.complex_shape()), (self._c_complex_shape, self.complex_shape())
        self._mesh = self._block(half_axis=2, drop_axes=1)
        self._create_plan()

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

    # -- host-side mesh helpers (line.py:105-134), answered by _mesh.Block from the layout --------------------------
    def get_local_mesh(self):
        """(2, *real_shape()) array of coordinates."""
        return self._mesh.coordinates_dense(self.float)

    def get_local_wavenumbermesh(self, scaled=True, broadcast=False, eliminate_highest_freq=False):
        """[Kx, Ky] of this rank's spectral block in double precision whatever the class's (as upstream); note the
        default `scaled=True`, unlike the 3-D classes."""
        return self._mesh.wavenumber_grid(dtype=np.float64, factors=2 * np.pi / self.L if scaled is True else None,
                                          cast_first=True, zero_nyquist=eliminate_highest_freq,
                                          dense=broadcast is True)

    def get_dealias_filter(self):
        """2/3-rule mask; upstream tests the SCALED wave numbers against the integer bound (line.py:131-134) -- kept."""
        return self._mesh.two_thirds_filter(self.get_local_wavenumbermesh())

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
