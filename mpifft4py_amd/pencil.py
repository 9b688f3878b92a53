"""Pencil decomposition on MI355X: counterparts of mpiFFT4py/pencil.py classes
R2CY (pencil.py:145-883), R2CX (pencil.py:885-1477) and the R2C factory
(pencil.py:1479-1484).

Real data (N0/P1, N1/P2, N2); complex data aligned in y (R2CY:
(N0/P2, N1, N1f)) or in x (R2CX: (N0, N1/P1, N2f)).  comm0 = P1 consecutive
world ranks, comm1 = P2 ranks of stride P1 (pencil.py:192-195).
"""
import numpy as np

from . import _lib, _padding
from ._base import DistFFTBase, default_planner_effort
from .comm import SubComm

__all__ = ["R2C", "R2CX", "R2CY", "C2C", "C2CX", "C2CY"]


def _compute_dims(nprocs):
    """MPI.Compute_dims(nprocs, 2): balanced, non-increasing."""
    best = (nprocs, 1)
    a = 1
    while a * a <= nprocs:
        if nprocs % a == 0:
            best = (nprocs // a, a)
        a += 1
    return best


class R2CY(DistFFTBase):
    """Pencil R2C with the final complex data aligned in the y-direction.

    `allow_single=True` lifts the reference's P > 1 assertion (pencil.py:176) so
    that the degenerate 1x1 grid can serve as the single-GPU scaling baseline."""
    _decomp = _lib.PENCIL_Y
    _kind = _lib.R2C

    def __init__(self, N, L, comm, precision, P1=None, communication='Alltoallw', padsize=1.5, threads=1,
                 planner_effort=None, allow_single=False, pipeline=0, allow_odd_grid=False, comm_cus=0, complex_pitch=None):
        self._comm_cus = comm_cus
        self._complex_pitch_req = complex_pitch       # see slab.R2C; pencil plans convert at the boundary
        self._init_common(N, L, comm, precision, communication, padsize, threads,
                          planner_effort if planner_effort is not None else default_planner_effort())
        N = self.N
        self.Nf = int(N[2] // 2 + 1)
        self.L = np.asarray(L).astype(self.float)      # upstream's `float` is the class dtype there (pencil.py:174-177)
        P = self.num_processes
        if not allow_single:
            assert P > 1
        if communication not in ('Alltoall', 'Alltoallw', 'AlltoallN'):
            raise ValueError("unknown communication %r" % (communication,))
        if P1 is None:
            P1, P2 = _compute_dims(P)
        else:
            P2 = P // P1
        self.P1, self.P2 = P1, P2
        # the reference's grid rules (pencil.py:202-208); `allow_odd_grid=True` lifts them: the C ABI takes any P1 that
        # divides P (1 x P and P x 1 grids are what an odd number of GPUs, or one exchange, would use)
        if not allow_odd_grid:
            if not (P % 2 == 0 or P == 1):
                raise IOError("Number of cpus must be even")
            if P > 1 and ((P1 % 2 != 0) or (P2 % 2 != 0)):
                raise IOError("Number of cpus in each direction must be even power of 2")
        self.N1 = N // P1
        self.N2 = N // P2
        self.comm0_rank = self.rank % P1
        self.comm1_rank = self.rank // P1
        self.comm0 = SubComm(P1, self.comm0_rank)
        self.comm1 = SubComm(P2, self.comm1_rank)
        if self._kind == _lib.R2C:
            self.N1f = int(self.N1[2] // 2) if self.comm0_rank < P1 - 1 else int(self.N1[2] // 2 + 1)
            self.N2f = int(self.N2[2] // 2) if self.comm1_rank < P2 - 1 else int(self.N2[2] // 2 + 1)
        else:                         # complex z axis: N2 columns, split evenly
            self.Nf = int(N[2])
            self.N1f = int(self.N1[2])
            self.N2f = int(self.N2[2])
        drop = communication == 'AlltoallN'
        if drop:
            # the z-Nyquist mode is neglected so that all chunks are equal (pencil.py:198-199, 909-910)
            assert self._kind == _lib.R2C
            self.N1f = int(self.N1[2] // 2)
            self.N2f = int(self.N2[2] // 2)
        self._describe(self._kind, self._decomp, p1=P1, pipeline=pipeline, drop_nyquist=drop)
        assert self._c_real_shape == tuple(self.real_shape())
        assert self._c_complex_shape == tuple(self.complex_shape()), (self._c_complex_shape, self.complex_shape())
        self._mesh = self._block(half_axis=2 if self._kind == _lib.R2C else None)
        self._create_plan()

    # -- shapes (pencil.py:248-287) ------------------------------------------------
    def real_shape(self):
        return (int(self.N1[0]), int(self.N2[1]), int(self.N[2]))

    def complex_shape(self):
        return (int(self.N2[0]), int(self.N[1]), self.N1f)

    def real_shape_padded(self):
        return (int(self.padsize * self.N1[0]), int(self.padsize * self.N2[1]), int(self.padsize * self.N[2]))

    def work_shape(self, dealias):
        return self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()

    def global_real_shape(self):
        return (int(self.N[0]), int(self.N[1]), int(self.N[2]))

    def global_complex_shape(self, padsize=1.0):
        return (int(padsize * self.N[0]), int(padsize * self.N[1]), int(padsize * self.N[2] // 2 + 1))

    def real_local_slice(self, padsize=1):
        c0, c1 = self.comm0_rank, self.comm1_rank
        return (slice(int(padsize * c0 * self.N1[0]), int(padsize * (c0 + 1) * self.N1[0]), 1),
                slice(int(padsize * c1 * self.N2[1]), int(padsize * (c1 + 1) * self.N2[1]), 1),
                slice(0, int(padsize * self.N[2])))

    def complex_local_slice(self):
        c0, c1 = self.comm0_rank, self.comm1_rank
        z0 = int(c0 * self.N1[2] // 2) if self._kind == _lib.R2C else int(c0 * self.N1[2])
        return (slice(int(c1 * self.N2[0]), int((c1 + 1) * self.N2[0]), 1),
                slice(0, int(self.N[1])),
                slice(z0, z0 + self.N1f, 1))

    def get_P(self):
        return self.P1, self.P2

    # host-side numpy helpers of the reference's API (pencil.py:351-379); the device path fuses these copies
    def copy_to_padded_x(self, fu, fp):
        return _padding.spread(fu, fp, self.N[0], 0)

    def copy_to_padded_y(self, fu, fp):
        return _padding.spread(fu, fp, self.N[1], 1)

    def copy_to_padded_z(self, fu, fp):
        fp[:, :, :self.Nf] = fu
        return fp

    def copy_from_padded_z(self, fp, fu):
        fu[:] = fp[:, :, :self.Nf]
        return fu

    def copy_from_padded_x(self, fp, fu):
        fu.fill(0)
        return _padding.gather_fold(fp, fu, self.N[0], 0)

    def copy_from_padded_y(self, fp, fu):
        fu.fill(0)
        return _padding.gather_fold(fp, fu, self.N[1], 1)

    # -- host-side mesh helpers (pencil.py:289-349, 945-969), answered by _mesh.Block from the layout ---------------
    # One contract for both alignments (upstream's x-aligned class returns ky unsliced from
    # `complex_local_wavenumbers` and has a kwarg-less `get_local_wavenumbermesh` that mis-slices kz on the ranks that
    # hold the Nyquist column; neither is reproduced).
    def complex_local_wavenumbers(self):
        """Integer (kx, ky, kz) vectors of this rank's spectral block."""
        return tuple(self._mesh.mode_vectors())

    def get_local_mesh(self):
        """[x, y, z] of this rank's pencil as read-only views of real_shape() (y-aligned class, pencil.py:299-312)."""
        return self._mesh.coordinates_sparse(self.float)

    def get_local_wavenumbermesh(self, scaled=False, broadcast=False, eliminate_highest_freq=False):
        """[Kx, Ky, Kz] of this rank's spectral block: integers, or -- `scaled` by 2 pi / L -- the class's real dtype;
        open unless `broadcast`; `eliminate_highest_freq` reports the Nyquist modes as zero."""
        return self._mesh.wavenumber_grid(dtype=self.float, factors=2 * np.pi / self.L if scaled is True else None,
                                          cast_first=False, zero_nyquist=eliminate_highest_freq,
                                          dense=broadcast is True)

    def get_dealias_filter(self):
        """The 2/3-rule mask of this rank's spectral block (uint8)."""
        return self._mesh.two_thirds_filter()

    # -- transforms ----------------------------------------------------------------
    def get_subarrays(self, padsize=1):
        """The Alltoallw boxes of the reference (pencil.py:218-246 for Y, 971-999 for X) as `Subarray` descriptors."""
        from . import _subarrays
        fn = _subarrays.pencil_x_subarrays if self._decomp == _lib.PENCIL_X else _subarrays.pencil_y_subarrays
        return fn([int(x) for x in self.N], self.Nf, self.P1, self.P2, self.comm0_rank, self.comm1_rank, padsize)

    def fftn(self, u, fu, dealias=None):
        """Forward transform (pencil.py:634-883 / 1228-1475); returns fu."""
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        ushape = self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()
        assert tuple(u.shape) == ushape
        return self._run(True, u, fu, dealias, ushape, self._in_dtype(), self.complex_shape(), self.complex)

    def ifftn(self, fu, u, dealias=None):
        """Inverse transform (pencil.py:386-632 / 1001-1224); fu is not modified."""
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        ushape = self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()
        assert tuple(u.shape) == ushape
        return self._run(False, fu, u, dealias, self.complex_shape(), self.complex, ushape, self._in_dtype())

    fft3d = fftn
    ifft3d = ifftn

    def _in_dtype(self):
        return self.float if self._kind == _lib.R2C else self.complex


class R2CX(R2CY):
    """Pencil R2C with the final complex data aligned in the x-direction."""
    _decomp = _lib.PENCIL_X

    def __init__(self, N, L, comm, precision, P1=None, communication='Alltoall', padsize=1.5, threads=1,
                 planner_effort=None, allow_single=False, pipeline=0, allow_odd_grid=False, comm_cus=0, complex_pitch=None):
        R2CY.__init__(self, N, L, comm, precision, P1=P1, communication=communication, padsize=padsize,
                      threads=threads, planner_effort=planner_effort, allow_single=allow_single,
                      pipeline=pipeline, allow_odd_grid=allow_odd_grid, comm_cus=comm_cus, complex_pitch=complex_pitch)

    def complex_shape(self):
        return (int(self.N[0]), int(self.N1[1]), self.N2f)

    def get_local_mesh(self):
        """(3, *real_shape()) array of coordinates (the x-aligned class returns a dense mesh, pencil.py:945-957)."""
        return self._mesh.coordinates_dense(self.float)

    def complex_local_slice(self):
        c0, c1 = self.comm0_rank, self.comm1_rank
        z0 = int(c1 * self.N2[2] // 2) if self._kind == _lib.R2C else int(c1 * self.N2[2])
        return (slice(0, int(self.N[0])),
                slice(int(c0 * self.N1[1]), int((c0 + 1) * self.N1[1]), 1),
                slice(z0, z0 + self.N2f, 1))


class C2CX(R2CX):
    """EXTENSION (no counterpart in the reference, which has no pencil C2C; asked for by
    BASELINE.json config "2048^3 fp32 complex-to-complex pencil"): complex <-> complex
    pencil transform, same layouts as R2CX with the full N2 columns split evenly.
    original (N0/P1, N1/P2, N2) <-> transformed (N0, N1/P1, N2/P2); un-padded only."""
    _kind = _lib.C2C

    def global_shape(self, padsize=1.):
        return (int(padsize * self.N[0]), int(padsize * self.N[1]), int(padsize * self.N[2]))

    original_shape = R2CY.real_shape
    original_local_slice = R2CY.real_local_slice

    def transformed_shape(self):
        return self.complex_shape()

    def transformed_local_slice(self):
        return self.complex_local_slice()


class C2CY(R2CY):
    """EXTENSION: complex <-> complex pencil transform aligned in y; transformed shape
    (N0/P2, N1, N2/P1)."""
    _kind = _lib.C2C

    def global_shape(self, padsize=1.):
        return (int(padsize * self.N[0]), int(padsize * self.N[1]), int(padsize * self.N[2]))

    original_shape = R2CY.real_shape
    original_local_slice = R2CY.real_local_slice

    def transformed_shape(self):
        return self.complex_shape()

    def transformed_local_slice(self):
        return self.complex_local_slice()


def C2C(N, L, comm, precision, P1=None, communication="Alltoallw", padsize=1.5, threads=1,
        alignment="X", planner_effort=None, **kw):
    cls = C2CX if alignment == 'X' else C2CY
    return cls(N, L, comm, precision, P1, communication, padsize, threads, planner_effort, **kw)


def R2C(N, L, comm, precision, P1=None, communication="Alltoall", padsize=1.5, threads=1,
        alignment="X", planner_effort=None, **kw):
    """Factory with the reference's signature (pencil.py:1479-1484)."""
    if alignment == 'X':
        return R2CX(N, L, comm, precision, P1, communication, padsize, threads, planner_effort, **kw)
    return R2CY(N, L, comm, precision, P1, communication, padsize, threads, planner_effort, **kw)
