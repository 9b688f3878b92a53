"""Slab decomposition on MI355X: drop-in counterparts of mpiFFT4py/slab.py
classes R2C (slab.py:49-536) and C2C (slab.py:538-825).

Real data (N0/P, N1, N2) <-> complex data (N0, N1/P, N2/2+1).  The transforms
themselves run in libmpifft4py_amd.so (HIP kernels + RCCL all-to-all); this
file keeps the reference's constructor signature, attributes, shape/slice
methods, mesh helpers and exception types.
"""
import numpy as np

from . import _lib, _padding
from ._base import DistFFTBase, default_planner_effort
from ._mesh import dft_modes

__all__ = ["R2C", "C2C"]


class R2C(DistFFTBase):
    """3-D real <-> complex FFT, slab decomposition.

    Args (as the reference, slab.py:67-71):
        N, L, comm, precision ("single"/"double"),
        communication ('Alltoall' | 'Alltoallw': identical results, both map to
        the RCCL exchange), padsize, threads, planner_effort (accepted, unused).
    Extension: pipeline -- the exchange of a multi-rank plan is cut into pieces that travel on a second stream
        while the next piece is transformed: n > 1 = n kz slices, n < -1 = |n| batches of local x rows,
        0 = default (4 kz slices), 1 = one blocking exchange.
        comm_cus -- pipelined plans: compute units set aside for the communication stream (0 = library default,
        < 0 = none), see include/mpifft4py_amd.h.
        complex_pitch -- None: the device spectrum is compact, rows of Nf bins (the reference's layout, slab.py:102-104);
        "auto": its rows lie a whole number of 128-byte cache lines apart (513 -> 520 bins), an integer: that many
        elements.  complex_shape() stays (N0, N1/P, Nf); allocate with FFT.empty_complex().  On one rank every pass
        then meets line-aligned rows; on several ranks the plan converts at the boundary.
    """
    _kind = _lib.R2C

    def __init__(self, N, L, comm, precision, communication="Alltoallw", padsize=1.5, threads=1,
                 planner_effort=None, pipeline=0, comm_cus=0, complex_pitch=None):
        self._comm_cus = comm_cus
        self._complex_pitch_req = complex_pitch
        self._init_common(N, L, comm, precision, communication, padsize, threads,
                          planner_effort if planner_effort is not None else default_planner_effort())
        N = self.N
        self.Nf = int(N[2] // 2 + 1)
        self.Nfp = int(padsize * N[2] // 2 + 1)
        self.Np = N // self.num_processes
        self.L = np.asarray(L).astype(self.float)
        if communication == "Sendrecv_replace":
            # dead code upstream (py2 xrange, shape bug for every P > 1); not offered
            raise ValueError("communication='Sendrecv_replace' is not supported")
        if communication not in ("Alltoall", "Alltoallw"):
            raise ValueError("unknown communication %r" % (communication,))
        if self.num_processes not in [2 ** i for i in range(int(np.log2(N[0])) + 1)]:
            raise IOError("Number of cpus must be in ", [2 ** i for i in range(int(np.log2(N[0])) + 1)])
        self._post_init()
        self._describe(self._kind, _lib.SLAB, pipeline=pipeline)
        assert self._c_real_shape == tuple(self.real_shape())
        assert self._c_complex_shape == tuple(self.complex_shape())
        self._mesh = self._block(half_axis=2 if self._kind == _lib.R2C else None)
        self._create_plan()

    def _post_init(self):
        pass

    # -- shapes (slab.py:98-144, 487-514) ---------------------------------------
    def real_shape(self):
        return (int(self.Np[0]), int(self.N[1]), int(self.N[2]))

    def complex_shape(self):
        return (int(self.N[0]), int(self.Np[1]), self.Nf)

    def complex_shape_T(self):
        return (int(self.Np[0]), int(self.N[1]), self.Nf)

    def global_real_shape(self):
        return (int(self.N[0]), int(self.N[1]), int(self.N[2]))

    def global_complex_shape(self, padsize=1.):
        return (int(padsize * self.N[0]), int(padsize * self.N[1]), int(padsize * self.N[2] // 2 + 1))

    def work_shape(self, dealias):
        return self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()

    def real_shape_padded(self):
        return (int(self.padsize * self.Np[0]), int(self.padsize * self.N[1]), int(self.padsize * self.N[2]))

    def complex_shape_padded_0(self):
        return (int(self.padsize * self.N[0]), int(self.Np[1]), self.Nf)

    def complex_shape_padded_1(self):
        return (int(self.padsize * self.Np[0]), int(self.N[1]), self.Nf)

    def complex_shape_padded_2(self):
        return (int(self.padsize * self.Np[0]), int(self.padsize * self.N[1]), self.Nf)

    def complex_shape_padded_3(self):
        return (int(self.padsize * self.Np[0]), int(self.padsize * self.N[1]), self.Nfp)

    def complex_shape_padded_0_I(self):
        return (self.num_processes, int(self.padsize * self.Np[0]), int(self.Np[1]), self.Nf)

    def complex_shape_padded_I(self):
        return (int(self.padsize * self.Np[0]), self.num_processes, int(self.Np[1]), self.Nf)

    # host-side numpy helpers of the reference's API (slab.py:516-536); the device path fuses these copies
    copy_to_padded = staticmethod(_padding.r2c_copy_to_padded)
    copy_from_padded = staticmethod(_padding.r2c_copy_from_padded)

    def real_local_slice(self, padsize=1):
        return (slice(int(padsize * self.rank * self.Np[0]), int(padsize * (self.rank + 1) * self.Np[0]), 1),
                slice(0, int(padsize * self.N[1]), 1),
                slice(0, int(padsize * self.N[2]), 1))

    def complex_local_slice(self):
        return (slice(0, int(self.N[0]), 1),
                slice(int(self.rank * self.Np[1]), int((self.rank + 1) * self.Np[1]), 1),
                slice(0, self.Nf, 1))

    # -- host-side mesh helpers (slab.py:146-197), answered by _mesh.Block from the layout --------------------------
    def complex_local_wavenumbers(self):
        """(kx, ky of this rank's columns, kz) as vectors of the class's real dtype."""
        return tuple(k.astype(self.float) for k in self._mesh.mode_vectors())

    def get_local_mesh(self):
        """[x, y, z] of this rank's planes, each a read-only view of real_shape()."""
        return self._mesh.coordinates_sparse(self.float)

    def get_local_wavenumbermesh(self, scaled=False, broadcast=False, eliminate_highest_freq=False):
        """[Kx, Ky, Kz] of this rank's spectral block in the class's real dtype: open (1-D along their own axis) unless
        `broadcast`; `scaled` by 2 pi / L; `eliminate_highest_freq` reports the Nyquist modes as zero."""
        return self._mesh.wavenumber_grid(dtype=self.float, factors=2 * np.pi / self.L if scaled else None,
                                          cast_first=True, zero_nyquist=eliminate_highest_freq,
                                          dense=broadcast is True)

    def get_dealias_filter(self):
        """The 2/3-rule mask of this rank's spectral block (uint8)."""
        return self._mesh.two_thirds_filter()

    # -- transforms ---------------------------------------------------------------
    def get_subarrays(self, padsize=1):
        """Subarrays for Alltoallw transforms (slab.py:199-211): the boxes every peer's chunk occupies in the send /
        receive arrays, as `Subarray(sizes, subsizes, starts)` descriptors instead of committed MPI datatypes."""
        from . import _subarrays
        return _subarrays.slab_subarrays([int(x) for x in self.N], [int(x) for x in self.Np], self.Nf, self.num_processes, padsize)

    def fftn(self, u, fu, dealias=None):
        """Forward transform (slab.py:349-485).  u: real_shape() (or
        real_shape_padded() with dealias='3/2-rule'); fu: complex_shape().
        numpy arrays or DeviceArrays; returns fu."""
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        ushape = self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()
        assert tuple(u.shape) == ushape
        return self._run(True, u, fu, dealias, ushape, self._in_dtype(), self.complex_shape(), self.complex)

    def ifftn(self, fu, u, dealias=None):
        """Inverse transform (slab.py:214-346); fu is not modified."""
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        ushape = self.real_shape_padded() if dealias == '3/2-rule' else self.real_shape()
        assert tuple(u.shape) == ushape
        return self._run(False, fu, u, dealias, self.complex_shape(), self.complex, ushape, self._in_dtype())

    # aliases named by the build brief
    fft3d = fftn
    ifft3d = ifftn

    def _in_dtype(self):
        return self.float


class C2C(R2C):
    """3-D complex <-> complex FFT, slab decomposition (slab.py:538-825)."""
    _kind = _lib.C2C

    def __init__(self, N, L, comm, precision, communication="Alltoall", padsize=1.5, threads=1,
                 planner_effort=None, pipeline=0, comm_cus=0, complex_pitch=None):
        R2C.__init__(self, N, L, comm, precision, communication=communication, padsize=padsize,
                     threads=threads, planner_effort=planner_effort, pipeline=pipeline, comm_cus=comm_cus,
                     complex_pitch=complex_pitch)

    copy_to_padded = staticmethod(_padding.c2c_copy_to_padded)        # slab.py:803-825
    copy_from_padded = staticmethod(_padding.c2c_copy_from_padded)

    def _post_init(self):
        N = self.N
        self.Nf = int(N[2])
        self.Nfp = int(self.padsize * N[2])
        self.original_shape_padded = self.real_shape_padded
        self.original_shape = self.real_shape
        self.transformed_shape = self.complex_shape
        self.original_local_slice = self.real_local_slice
        self.transformed_local_slice = self.complex_local_slice
        self.ks = dft_modes(N[2])      # exact integers (the reference truncates a float: see oracle._exact_ks)

    def global_shape(self, padsize=1.):
        return (int(padsize * self.N[0]), int(padsize * self.N[1]), int(padsize * self.N[2]))

    def transformed_local_wavenumbers(self):
        """Float64 wave vectors of the transformed block (slab.py:612-616)."""
        return tuple(k.astype(np.float64) for k in self._mesh.mode_vectors())

    def _in_dtype(self):
        return self.complex
