"""Counterpart of mpiFFT4py/mpibase.py: dtypes per precision, host array
helpers and the `work_arrays` cache that the reference exports and its demo
uses (demo/spectral_dns_solver.py:19,42,67-68).

Semantics kept from the reference (mpibase.py:61-131): keys are
(shape, dtype, index[, fillzero]) or (ndarray, index[, fillzero]); an array is
created on first use and zero-filled on every access unless fillzero is False.
"""
import collections.abc

import numpy as np

try:                      # only used to hand the matching MPI datatype back to callers
    from mpi4py import MPI as _MPI
except Exception:         # mpi4py is optional
    _MPI = None


def empty(N, dtype=float, bytes=None):
    return np.empty(N, dtype=dtype)


def zeros(N, dtype=float, bytes=None):
    return np.zeros(N, dtype=dtype)


class work_arrays(collections.abc.MutableMapping):
    def __init__(self):
        self.store = {}
        self.fillzero = True

    def _key(self, key):
        if isinstance(key[0], np.ndarray):
            shape, dtype, i = key[0].shape, key[0].dtype, key[1]
            zero = True if len(key) == 2 else key[2]
        elif isinstance(key[0], tuple):
            if len(key) == 3:
                shape, dtype, i = key
                zero = True
            elif len(key) == 4:
                shape, dtype, i, zero = key
            else:
                raise TypeError("Wrong type of key for work array")
        else:
            raise TypeError("Wrong type of key for work array")
        assert isinstance(zero, bool)
        assert isinstance(i, int)
        self.fillzero = zero
        return (tuple(shape), np.dtype(dtype), i)

    def __getitem__(self, key):
        k = self._key(key)
        if k not in self.store:
            self.store[k] = np.zeros(k[0], dtype=k[1])
        val = self.store[k]
        if self.fillzero is True:
            val.fill(0)
        return val

    def __setitem__(self, key, value):
        self.store[self._key(key)] = value

    def __delitem__(self, key):
        del self.store[self._key(key)]

    def __iter__(self):
        return iter(self.store)

    def __len__(self):
        return len(self.store)

    def values(self):
        raise TypeError("Work arrays not iterable")


def datatypes(precision):
    """(real dtype, complex dtype, MPI datatype or name) for 'single'/'double'
    (mpibase.py:133-137)."""
    assert precision in ("single", "double")
    if precision == "single":
        return (np.float32, np.complex64, _MPI.C_FLOAT_COMPLEX if _MPI else "C_FLOAT_COMPLEX")
    return (np.float64, np.complex128, _MPI.C_DOUBLE_COMPLEX if _MPI else "C_DOUBLE_COMPLEX")
